// Write-through stores for data whose only reader is the NEXT kernel of the layer (sampled batches: Q|K|V, logits, the run rows).
// A plain store leaves its line dirty in the writing XCD's L2 and the whole lot is written back at the kernel boundary -- the
// successor starts B / 6 TB/s later (MI355X_MICROARCH.md, "boundary": 2.8 - 3.8 us behind 12.6 - 16.8 MB); an `sc1` store goes out while
// the kernel still computes.  16-byte stores only: narrower sc1 stores are one fabric write each (dword ~6x the time per byte of
// dwordx4) -- the logits / run statistics (a megabyte at these sizes) stay plain.  HGT_WT_STORES=0 compiles the plain form (A/B builds).
#pragma once
#include <hip/hip_runtime.h>
#ifndef HGT_WT_STORES
#define HGT_WT_STORES 1
#endif
namespace {
typedef float hgt_wt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt16(float* p, float a, float b, float c, float d) {
#if HGT_WT_STORES
    const hgt_wt_f4 v = {a, b, c, d};
    // (s_nop: a store of more than 64 bits followed by a VALU write of its data registers needs wait states hipcc inserts for its own
    //  stores and cannot insert behind inline asm -- without them the first build of this header published corrupted rows)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}
}  // namespace
