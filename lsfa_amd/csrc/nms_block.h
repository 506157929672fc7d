// Greedy NMS inside one 64-candidate block, for a single wave (shared by nms_kernels.h and detpost.hip).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace lsfa {

constexpr int kFixpointTries = 3;

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src_lane) {
  const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
  const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
  return ((uint64_t)hi << 32) | lo;
}

// Which candidates of a block survive, given for each lane (= candidate, in score order):
//   alive  the candidate exists and no survivor of an earlier block suppresses it; cand = __ballot(alive)
//   colw   bit j set <=> candidate j < lane of this block suppresses this lane   (transposed diagonal word)
//   rowd   bit j set <=> this lane suppresses candidate j > lane of this block   (diagonal word of its mask row)
// Greedy rule: k survives <=> alive(k) and no surviving j < k suppresses k.  Stops after `budget` survivors
// (the first `budget` in order).  Returns the survivor set, wave-uniform.
//
// Two ways to the same set.  Fixpoint: G <- {k alive : colw_k & G == 0}, starting from all alive; it alternates
// between over- and under-estimates and is exact once it repeats, after as many steps as the longest chain
// "a suppresses b suppresses c ..." — 2-3 steps when boxes overlap little, ~60 when a block is one pile of
// near-duplicates.  Scan: take the first candidate, strike what it suppresses, repeat — one scalar step per
// SURVIVOR, so the pile costs a handful of steps.  The fixpoint gets kFixpointTries steps, then the scan takes over.
__device__ __forceinline__ uint64_t resolve_block(uint64_t cand, bool alive, uint64_t colw, uint64_t rowd, int budget) {
  if (cand == 0 || budget <= 0) return 0;
  uint64_t G = cand;
  bool settled = false;
  for (int it = 0; it < kFixpointTries; ++it) {
    const uint64_t G2 = __ballot(alive && (colw & G) == 0);
    if (G2 == G) { settled = true; break; }
    G = G2;
  }
  if (!settled) {
    G = 0;
    uint64_t left = cand;
    int room = budget;
    while (left && room > 0) {
      const int k = __builtin_ctzll(left);
      G |= 1ULL << k;
      --room;
      left &= ~(readlane64(rowd, k) | (1ULL << k));
    }
    return G;
  }
  if (__popcll(G) > budget) {      // keep the first `budget` survivors
    const int lane = threadIdx.x & 63;
    const bool mine = (G >> lane) & 1ULL;
    const int rank = __popcll(G & ((1ULL << lane) - 1ULL));
    G = __ballot(mine && rank < budget);
  }
  return G;
}

}  // namespace lsfa
