// sortscan.h -- the two device-wide primitives of the preparation ops (grid.hip, volume.hip), hand-written for wave64 / LDS:
// an exclusive prefix sum of 32-bit counts and a stable LSD radix sort of (64-bit key, 32-bit value) pairs.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ps {

// workspace, in unsigned words, of exclusive_scan_u32 over n elements
size_t scan_workspace_words(size_t n);
// out[i] = in[0] + ... + in[i-1] (mod 2^32); in and out may be the same array.  Enqueues on st, never synchronises.
void exclusive_scan_u32(hipStream_t st, const unsigned* in, unsigned* out, size_t n, unsigned* work);

// workspace, in unsigned words, of radix_sort_pairs_u64 over n pairs
size_t sort_workspace_words(size_t n);
// Stable sort of (key, value) by the key's low `bits` bits (keys must be < 2^bits), eight bits per pass, ping-pong between
// (k0, v0) = input and (k1, v1).  Returns 0 / 1: which pair of arrays holds the sorted result.
int radix_sort_pairs_u64(hipStream_t st, unsigned long long* k0, unsigned long long* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work);
// the same with 32-bit keys (bits <= 32): a third less traffic per pass
int radix_sort_pairs_u32(hipStream_t st, unsigned* k0, unsigned* k1, unsigned* v0, unsigned* v1, size_t n, int bits, unsigned* work);

}  // namespace ps
