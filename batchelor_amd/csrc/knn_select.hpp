// Selection machinery shared by the MFMA top-k kernels (f32 and split-bf16 candidate passes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bmx {
namespace sel {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t f32_orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}

// Per query slot the LDS holds KS "kept" entries (sorted, shared by the two lanes that own the query's two K-halves)
// followed by two lane-private pending lists of PL entries each: a lane appends with a plain ds_write (no atomics, no
// returned value to wait for).  When a pending list fills, one wave merges kept + both pending lists by rank-counting
// over the unique 64-bit keys (orderable value << 32 | reference index), keeps the KS smallest and tightens tau.
constexpr int PL = 12;
// Row pitch of a query's list in 8-byte entries (register-state variant): one more than the capacity, so that the
// lists of neighbouring queries start 2 banks apart and the appends of a wave's lanes do not pile up on one bank pair.
__host__ __device__ constexpr int list_pitch(int KS, int PLN) { return KS + 2 * PLN + 1; }

template <int KS>
__device__ __forceinline__ void compact_slot(unsigned long long* buf, int* kcnt, float* tau_s, int slot, int jj,
                                             int lane, int& mycnt) {
    constexpr int CAP = KS + 2 * PL;
    const int nk = kcnt[slot];
    const int n0 = __builtin_amdgcn_readlane(mycnt, jj);
    const int n1 = __builtin_amdgcn_readlane(mycnt, jj + 32);
    const int n = nk + n0 + n1;
    unsigned long long* b = buf + slot * CAP;
    int src = lane;  // kept entries sit at [0, nk)
    if (lane >= nk) src = lane < nk + n0 ? KS + (lane - nk) : KS + PL + (lane - nk - n0);
    const unsigned long long key = lane < n ? b[src] : ~0ull;
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    int rank = 0;
    for (int f = 0; f < n; ++f) {
        const uint32_t flo = __builtin_amdgcn_readlane(klo, f);
        const uint32_t fhi = __builtin_amdgcn_readlane(khi, f);
        const unsigned long long fk = ((unsigned long long)fhi << 32) | flo;
        rank += fk < key ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its key before slots are rewritten
    if (lane < n && rank < KS) b[rank] = key;
    if (n >= KS && lane < n && rank == KS - 1) tau_s[slot] = orderable_f32(khi);
    if (lane == 0) kcnt[slot] = n < KS ? n : KS;
    if (lane == jj || lane == jj + 32) mycnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Register-state variant used by the split-bf16 ring kernel: the number of kept entries (nk) and the working
// threshold (tau) of a query live in the registers of the two lanes that own it, so a compaction makes ONE LDS round
// trip (the entries).  Entries sit in the LDS as (raw f32 bits << 32 | reference index).
// Ranking uses a 31-bit key per lane: the order-preserving integer image of the value, shifted right by one, with
// its low 6 bits replaced by the lane number -- unique, and a comparison is the sign of a difference; entries are
// broadcast by readlane in 8-entry chunks with constant lane numbers.
// Values closer than 2^-16 relative may therefore swap places at the cut; the new threshold is the cut key with the
// lane bits cleared, which is <= the value of everything dropped, so "rejected => value >= tau" still holds exactly.
template <int KS, int PLN>
__device__ __forceinline__ void compact_regs(unsigned long long* buf, int slot, int jj, int lane, int& mycnt, int& nk_reg,
                                             float& tau) {
    constexpr int CAP = KS + 2 * PLN;
    const int nk = __builtin_amdgcn_readlane(nk_reg, jj);
    const int n0 = __builtin_amdgcn_readlane(mycnt, jj);
    const int n1 = __builtin_amdgcn_readlane(mycnt, jj + 32);
    const int n = nk + n0 + n1;
    unsigned long long* b = buf + slot * list_pitch(KS, PLN);
    int src = lane;  // kept entries sit at [0, nk)
    if (lane >= nk) src = lane < nk + n0 ? KS + (lane - nk) : KS + PLN + (lane - nk - n0);
    const unsigned long long raw = lane < n ? b[src] : 0ull;
    const uint32_t key =
        lane < n ? (((f32_orderable(__uint_as_float((uint32_t)(raw >> 32))) >> 1) & ~63u) | (uint32_t)lane) : 0x7FFFFFFFu;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < (CAP + 7) / 8; ++c) {
        if (c * 8 < n) {
            // 31-bit keys: the sign of the difference is the comparison, so no lane mask (and no scalar-register
            // round trip per entry) is involved
            uint32_t fk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) fk[e] = c * 8 + e < CAP ? (uint32_t)__builtin_amdgcn_readlane(key, c * 8 + e) : 0x7FFFFFFFu;
#pragma unroll
            for (int e = 0; e < 8; ++e) rank += (fk[e] - key) >> 31;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its entry before slots are rewritten
    if (lane < n && rank < KS) b[rank] = raw;
    const bool mine = lane == jj || lane == jj + 32;
    if (n >= KS) {
        const unsigned long long at = __builtin_amdgcn_ballot_w64(lane < n && rank == KS - 1);
        const float kth = orderable_f32(((uint32_t)__builtin_amdgcn_readlane(key, __builtin_ctzll(at)) & ~63u) << 1);
        if (mine) tau = kth < tau ? kth : tau;
    }
    if (mine) {
        mycnt = 0;
        nk_reg = n < KS ? n : KS;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

}  // namespace sel
}  // namespace bmx
