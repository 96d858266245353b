#!/bin/bash
# Samples socket power and shader clock while the tower kernel runs back to back at full occupancy
# (1024 positions, 256 workgroups) and at half (512 positions, 128 workgroups, same kernel):
# evidence for the power limit behind the clock drop.  Reads sensors only.
set -u
sample() {
  local p c
  p=$(amd-smi metric -p 2>/dev/null | grep -m1 SOCKET_POWER | sed 's/.*SOCKET_POWER: *//')
  c=$(rocm-smi -c 2>/dev/null | grep -m1 sclk | sed 's/.*(\(.*\))/\1/')
  echo "socket power $p, sclk $c"
}
probe() {  # label, batch
  AZ_TOWER_FORM=4 python3 -u tools/net_bench.py --batch $2 --iters 14000 > /tmp/nb_$1.log 2>&1 &
  local pid=$!
  sleep 4
  for i in 1 2 3 4 5; do echo "$1 (batch $2): $(sample)"; sleep 0.4; done
  wait $pid
  grep batch /tmp/nb_$1.log
}
echo "power cap: $(rocm-smi --showmaxpower 2>/dev/null | grep -m1 -i 'Max Graphics Package Power' | sed 's/.*: //') W"
echo "idle: $(sample)"
probe full 1024
probe half 512
