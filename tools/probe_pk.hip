// Probe: is a packed-f32 VALU add bit-identical to two scalar adds on gfx950 when waves of ANOTHER workgroup on the same CU
// issue MFMAs?  (Diagnosis of the run-to-run different GroupNorm statistics of conv tile 1 -- the one conv kernel that keeps two
// workgroups per CU, so one workgroup's epilogue overlaps the other's MFMA main loop; DESIGN.md, Determinism.)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_pk.hip -o tools/_diag/probe_pk && tools/_diag/probe_pk
// Every probe lane accumulates the same pseudo-random stream twice: packed (form under test) and with scalar v_add_f32 /
// v_fmac_f32 in inline asm (never re-packed by the compiler), and counts the iterations after which the two differ.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// FORM 0: v_pk_add_f32 plain   1: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (halves of src1 swapped)   2: v_pk_fma_f32 plain
// FORM 3: scalar control (v_add_f32 both ways)   4: as 1, src1's high register produced by an SDWA shift (the epilogue's idiom)
template <int FORM>
__global__ void __launch_bounds__(256) probe(unsigned* bad_lo, unsigned* bad_hi, int iters, int mfma_mode, float* sink) {
    const int lane = threadIdx.x & 63;
    const bool is_mfma = (mfma_mode == 1 && (blockIdx.x & 1)) || (mfma_mode == 2);
    if (is_mfma && !(mfma_mode == 2 && (threadIdx.x >> 6) < 2)) {
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3c00 + lane * 3 + j); b[j] = (short)(0x3d00 + lane + 5 * j); }
        for (int i = 0; i < iters * 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[7];
        return;
    }
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    f32x2 s = {0.f, 0.f};
    float r0 = 0.f, r1 = 0.f;
    unsigned nlo = 0, nhi = 0;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const unsigned u0 = h & 0xffff0000u;                    // two bf16-like values in [-2, 2)
        h = h * 1664525u + 1013904223u;
        const unsigned u1 = h >> 16;
        float x0 = __uint_as_float((u0 & 0x807f0000u) | 0x3f800000u) - 1.5f;
        float x1;
        if (FORM == 4) {
            unsigned t = (u1 & 0x807fu) | 0x3f80u, sh = 16;
            asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(x1) : "v"(sh), "v"(t));
            asm volatile("v_add_f32 %0, -1.5, %0" : "+v"(x1));
        } else {
            x1 = __uint_as_float(((u1 << 16) & 0x807f0000u) | 0x3f800000u) - 1.5f;
        }
        f32x2 x = {x0, x1};
        if (FORM == 0) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s) : "v"(x));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(x0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(x1));
        } else if (FORM == 1 || FORM == 4) {
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(s) : "v"(x));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(x1));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(x0));
        } else if (FORM == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(s) : "v"(x));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r0) : "v"(x0));
            asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r1) : "v"(x1));
        } else {
            float a0 = s[0], a1 = s[1];
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(x0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(x1));
            s[0] = a0; s[1] = a1;
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r0) : "v"(x0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(r1) : "v"(x1));
        }
        if (__float_as_uint(s[0]) != __float_as_uint(r0)) { ++nlo; s[0] = r0; }
        if (__float_as_uint(s[1]) != __float_as_uint(r1)) { ++nhi; s[1] = r1; }
        if ((i & 63) == 63) { s[0] = r0 = 0.f; s[1] = r1 = 0.f; }
    }
    bad_lo[blockIdx.x * 256 + threadIdx.x] = nlo;
    bad_hi[blockIdx.x * 256 + threadIdx.x] = nhi;
}

template <int FORM>
static void run(const char* name, int mode, unsigned* dlo, unsigned* dhi, float* sink, int blocks, int iters) {
    CK(hipMemset(dlo, 0, blocks * 256 * 4));
    CK(hipMemset(dhi, 0, blocks * 256 * 4));
    probe<FORM><<<blocks, 256>>>(dlo, dhi, iters, mode, sink);
    CK(hipDeviceSynchronize());
    static unsigned hl[1 << 20], hh[1 << 20];
    CK(hipMemcpy(hl, dlo, blocks * 256 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hh, dhi, blocks * 256 * 4, hipMemcpyDeviceToHost));
    unsigned long long lo = 0, hi = 0, lanes = 0;
    for (int i = 0; i < blocks * 256; ++i) { lo += hl[i]; hi += hh[i]; lanes += (hl[i] | hh[i]) != 0; }
    printf("%-44s mfma_mode %d: low-half mismatches %llu, high-half %llu, lanes hit %llu of %d\n", name, mode, lo, hi, lanes, blocks * 256);
}

int main() {
    const int blocks = 2048, iters = 4096;
    unsigned *dlo, *dhi; float* sink;
    CK(hipMalloc(&dlo, blocks * 256 * 4)); CK(hipMalloc(&dhi, blocks * 256 * 4)); CK(hipMalloc(&sink, blocks * 256 * 4));
    for (int mode = 0; mode < 3; ++mode) {      // 0: probes only   1: odd workgroups run MFMAs   2: waves 2,3 of every workgroup run MFMAs
        run<3>("scalar control", mode, dlo, dhi, sink, blocks, iters);
        run<0>("v_pk_add_f32", mode, dlo, dhi, sink, blocks, iters);
        run<1>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", mode, dlo, dhi, sink, blocks, iters);
        run<4>("  same, src1.hi from an SDWA shift", mode, dlo, dhi, sink, blocks, iters);
        run<2>("v_pk_fma_f32", mode, dlo, dhi, sink, blocks, iters);
    }
    return 0;
}
