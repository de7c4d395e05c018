// Cross-workgroup hand-off inside one launch ("the last workgroup to arrive folds the others' partial rows"), shared by
// rollout.hip (rollout_pre / rollout_post) and rlg.hip (episode meters).
#pragma once

#include "common.h"

// Data that crosses workgroups inside a launch (partial rows -> the workgroup that folds them).  Device-scope FENCES are
// what this must not be built on: on gfx950 a __threadfence() is an L2 write-back / invalidate of the whole XCD slice and
// measured 10-25 us per launch when every workgroup executes one (profiles/r3_fna_variants.txt).  The partial rows are
// instead written with device-scope atomic stores and read with device-scope atomic loads (both go to the coherence
// point by themselves, nothing else needs flushing), ordered by waiting for the stores' completion (s_waitcnt vmcnt(0))
// before the arrive.  ROLLOUT_FENCES=1 rebuilds the fence version (A/B).
#ifndef ROLLOUT_FENCES
#define ROLLOUT_FENCES 0
#endif
// The fence-free form has no release / acquire in the HIP memory model: it relies on gfx9-family behaviour - stores are
// counted in vmcnt (gfx10+ count them in vscnt: the s_waitcnt below would not wait for them) and agent-scope atomic
// stores / loads (sc1) are served at the coherence point.  Any other target must build the fence version.
#if defined(__HIP_DEVICE_COMPILE__) && !ROLLOUT_FENCES && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "xwg.h: the fence-free cross-workgroup hand-off is written for the gfx9 family (gfx950); build with -DROLLOUT_FENCES=1"
#endif
template <typename T>
__device__ __forceinline__ void xwg_store(T* p, T v) {
#if ROLLOUT_FENCES
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
template <typename T>
__device__ __forceinline__ T xwg_load(const T* p) {
#if ROLLOUT_FENCES
  return __builtin_nontemporal_load(p);
#else
  return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ bool last_block_arrives(unsigned int* ticket, unsigned int expected) {
  // "last workgroup folds" hand-shake: every xwg_store of this workgroup has completed, take a ticket; the last one
  // reads the other workgroups' rows with xwg_load
  __shared__ int s_last;
#if ROLLOUT_FENCES
  __syncthreads();          // every store of the workgroup has left the CU (write-through L1) ...
  if (threadIdx.x == 0) {
    __threadfence();        // ... release at device scope: L2 write-back, visible to the other XCDs
    const unsigned int t = atomicAdd(ticket, 1u);
    s_last = (t == expected - 1) ? 1 : 0;
    if (s_last) *ticket = 0u;   // ready for the next launch (stream ordered)
  }
  __syncthreads();
  const bool last = s_last != 0;
  if (last) __threadfence();   // acquire: drop stale cache lines before reading the other workgroups' partials
  return last;
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's device-scope stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(ticket, 1u);
    s_last = (t == expected - 1) ? 1 : 0;
    if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch (stream ordered)
  }
  __syncthreads();
  return s_last != 0;
#endif
}
