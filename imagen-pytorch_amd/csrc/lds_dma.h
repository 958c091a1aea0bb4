// lds_dma.h — the gfx950 primitives of the direct-to-LDS conv kernels (conv_dma.hip, conv_stream.hip, conv_big.hip, igemm.hip's barrier) behind
// ONE switch.  The product build is the inline assembly below; -DIMAGEN_EMUL (tools/emul: the CPU functional emulation, test infrastructure)
// maps the same names onto the emulator's hooks — a copy is queued and lands at the covering vmcnt wait, in issue order; barriers are the
// fiber rendezvous.  The kernels themselves carry no #ifdef.  The copy macros expect the kernel's dynamic LDS array to be called `smem`.
#pragma once

#ifdef IMAGEN_EMUL
#define IMAGEN_DMA16(gsrc, lds_dst) emul::dma16(gsrc, lds_dst, smem)          // lane l -> LDS bytes [dst + 16 l, +16)
#define IMAGEN_DMA4(gsrc, lds_dst) emul::dma4(gsrc, lds_dst, smem)            // lane l -> LDS bytes [dst + 4 l, +4)
#define IMAGEN_WARM_DMA4(gsrc, lds_dst) ((void)(gsrc), (void)(lds_dst))       // (a cache warm-up into a sink slot: nothing to emulate)
#define IMAGEN_LDS_BASE(ptr) 0u
#define IMAGEN_BARRIER() __syncthreads()
#define IMAGEN_LGKM0_BARRIER() __syncthreads()
#define IMAGEN_VM0_BARRIER() do { emul::wait_vm(0); __syncthreads(); } while (0)
#define IMAGEN_WAIT_VM_STORES(n) ((void)(n), emul::wait_vm(0))   // (the allowance counts the lane's output STORES behind its copies; the emulation queues copies only)
#define IMAGEN_OPAQUE(v) ((void)0)
#define IMAGEN_SINK(v) ((void)(v))
#else
__device__ __forceinline__ void imagen_dma16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void imagen_dma4(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void imagen_wait_vm_le12(int n) {   // s_waitcnt vmcnt(min(n, 12)): fewer outstanding than allowed is always safe
  switch (n < 12 ? n : 12) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}
#define IMAGEN_DMA16(gsrc, lds_dst) imagen_dma16(gsrc, lds_dst)
#define IMAGEN_DMA4(gsrc, lds_dst) imagen_dma4(gsrc, lds_dst)
#define IMAGEN_WARM_DMA4(gsrc, lds_dst) imagen_dma4(gsrc, lds_dst)
#define IMAGEN_LDS_BASE(ptr) ((unsigned)(size_t)(__attribute__((address_space(3))) char*)(ptr))
#define IMAGEN_BARRIER() asm volatile("s_barrier" ::: "memory")
#define IMAGEN_LGKM0_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   // global loads / copies stay in flight across it
#define IMAGEN_VM0_BARRIER() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
#define IMAGEN_WAIT_VM_STORES(n) imagen_wait_vm_le12(n)
#define IMAGEN_OPAQUE(v) asm volatile("" : "+v"(v))      // makes a value opaque to the optimiser (nothing derived from it is hoisted)
#define IMAGEN_SINK(v) asm volatile("" ::"v"(v))         // "uses" a value (keeps the loads that produced it alive)
#endif
// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]) + a compiler-level fence; both builds
#define IMAGEN_WAIT_VM(n)                                                                     \
  do {                                                                                        \
    __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)); \
    asm volatile("" ::: "memory");                                                            \
  } while (0)
