// Library-internal C++ interface between the network (net.hip) and the fused
// self-play loop (selfplay.hip).
#pragma once
#include "head_dev.h"

struct az_net;

namespace az {
// Row assignment on the device: the rows of the network batch are the games whose flag in
// `need` is set, in game order.  The tower finds them itself (no host read, no extra
// launch), writes row_of[game] for each and the number of rows to *n_rows; workgroups and
// head tiles past that number leave at once.  `need` must be readable, as zeros, up to the
// next multiple of 16 games.
struct RowAssign {
    const uint8_t* need;
    int32_t n_games;
    int32_t* row_of;
    int32_t* n_rows;
    unsigned long long* rows_total;   // += rows of this launch (NULL: not counted)
};
// stem + tower + head GEMMs for up to B positions given as canonical bitboards; the
// split-K partial sums stay in the network's own buffers.  With rows == NULL position i is
// mine_dev[i] / theirs_dev[i]; with rows it is that of the i-th flagged game and B is the
// number of rows the launch is sized for (at least the number of flags that can be set).
int net_forward_partial(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                        const RowAssign* rows, hipStream_t s);
// the same followed by the head finish: probabilities [B][192] and values [B]
int net_forward_full(az_net* n, int32_t B, const uint64_t* mine_dev, const uint64_t* theirs_dev,
                     const RowAssign* rows, float* probs_dev, float* value_dev, float* policy_logits_dev,
                     float* wl_logits_dev, hipStream_t s);
// what dev_head_finish needs to turn those partial sums into probabilities / values
FinishParams net_finish_params(az_net* n, int32_t B, const int32_t* n_rows);
// HIP device ordinal the network's weights and buffers live on
int net_device(const az_net* n);
}  // namespace az
