// Hash of a BPE merge key (left id << 32 | right id); shared by the host table builder and the
// device lookup so both probe the same slots.
#pragma once
#include <stdint.h>
namespace czc {
__host__ __device__ inline unsigned bridge_hash(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (unsigned)k;
}
}  // namespace czc
