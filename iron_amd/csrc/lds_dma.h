// LDS-DMA (global_load_lds) issued from inline assembly.
//
// Why not __builtin_amdgcn_global_load_lds: hipcc 7.2's waitcnt insertion books an LDS-DMA as a FLAT access that may touch LDS, and
// while one is pending every wait it emits for a ds_read is the pessimistic s_waitcnt lgkmcnt(0) (and vmcnt(0) for global loads).
// In a weight ring there is ALWAYS a DMA pending, so each group of MFMAs waited for the fragments just requested for the NEXT group
// as well: software prefetch of A fragments was impossible (702 of 702 waits in k_sdf_values_h2 were lgkmcnt(0)).  From assembly
// the compiler does not see the DMA, counts only its own ds_reads and emits exact lgkmcnt(N).  The ring then owns the ordering of
// DMA writes against LDS reads by itself: s_waitcnt vmcnt(N) + s_barrier before a slot is read (Ring::sync_take), which it did
// already.  Extra younger VMEM operations of the compiler only make that vmcnt wait more conservative (in-order return).
#pragma once
#include <stdint.h>

namespace iron {

// 64 lanes x 16 B: global `src` (per-lane address) -> LDS [lds_base + lane * 16]
__device__ __forceinline__ void lds_dma16(const char* src, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory");
}
// 64 lanes x 4 B: -> LDS [lds_base + lane * 4]
__device__ __forceinline__ void lds_dma4(const char* src, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_base) : "memory");
}
// The same with the address split as the ISA's saddr form wants it: a wave-uniform 64-bit base in SGPRs + one 32-bit per-lane offset
// (lane * 16 or lane * 4, a constant VGPR).  A ring refill then costs the vector port nothing per piece: the base moves by scalar adds
// (the per-lane 64-bit pointer form needs one v_lshl_add_u64 per piece, 17 per ring step in k_sdf_values_h2).
__device__ __forceinline__ void lds_dma16_s(const char* base_uniform, uint32_t lane_off, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(base_uniform), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void lds_dma4_s(const char* base_uniform, uint32_t lane_off, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_off), "s"(base_uniform), "s"(lds_base) : "memory");
}
// wave-uniform LDS byte address of a (generic) pointer into __shared__ memory: the low half of the flat address IS the LDS offset
// (aperture base in the high half); an addrspacecast would add a null check, which hipcc 7.2 mis-selects in some kernels
// ("Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base")
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
}

}  // namespace iron
