// Sort stage of the fixed-base MSM with ONE bucket set and windows wider than 16 bits (msm_sort_wide.hip): 2^16 .. 2^21 buckets.
#pragma once
#include "msm_sort.hpp"

namespace csh {

// Host-side plan of the wide sort (everything the kernels need beyond MsmParams)
struct WidePlan {
  uint32_t lb;           // log2 of the buckets per partition (level 2 sorts by the low lb bits of bucket - 1): 8 .. 11
  uint32_t P;            // partitions = NB >> lb (level 1 sorts by the high bits)
  uint32_t B2;           // buckets per partition = 1 << lb
  uint32_t n_tiles;      // level-1 tiles: WS_BLK scalars x W digits each
  uint32_t chunk_tiles;  // tiles per level-1 block
  uint32_t CH;           // level-1 blocks (chunks of consecutive scalars)
  uint32_t jobs_max;     // upper bound of the level-2 jobs (one tile of one partition each)
  uint32_t rec4;         // 1: 4-byte intermediate records (every stored entry id fits 31 - lb bits)
};

// fr_id of the scalar field (Fr of the curve): 0 = BN254 Fr, 1 = BLS12-381 Fr, 2 = BN254 Fq (Grumpkin's scalars)
WidePlan msm_wide_plan(const MsmParams& srt, const MsmParams& dig);
size_t msm_sort_wide_bytes(const MsmParams& srt, const MsmParams& dig);
// srt: the merged plan's sort half (n = points x rows entries, W = 1, c in 17 .. 22, remap_*), dig: its digit half (n points, W rows).
// Takes its buffers from `ar` (already reserved); out_*: what the bucket stage consumes. ev (nullable): records ev[1] after level 1,
// ev[2] after the bucket histogram + scan, ev[3] after level 2.
int msm_sort_wide_launch(int fr_id, const MsmParams& srt, const MsmParams& dig, const uint64_t* scalars_dev, hipStream_t st, Arena& ar,
                         uint32_t** out_start, uint32_t** out_nlanes, uint32_t** out_sorted, hipEvent_t* ev);

}  // namespace csh
