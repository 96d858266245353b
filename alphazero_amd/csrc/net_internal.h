// Library-internal C++ interface between the network (net.hip) and the fused
// self-play loop (selfplay.hip).
#pragma once
#include "head_dev.h"

struct az_net;

namespace az {
// stem + tower + head GEMMs for B positions given as canonical bitboards; the
// split-K partial sums stay in the network's own buffers.  With gather_dev, position i
// is read from mine_dev[gather_dev[i]] / theirs_dev[gather_dev[i]].
int net_forward_partial(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                        const int32_t* gather_dev, hipStream_t s);
// the same followed by the head finish: probabilities [B][192] and values [B]
int net_forward_full(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                     const int32_t* gather_dev, float* probs_dev, float* value_dev, float* policy_logits_dev,
                     float* wl_logits_dev, hipStream_t s);
// what dev_head_finish needs to turn those partial sums into probabilities / values
FinishParams net_finish_params(az_net* n, int32_t B);
}  // namespace az
