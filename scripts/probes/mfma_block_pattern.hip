// What does each ingredient of chain3.hip's k-step block cost?  One wave per SIMD (waves 0-3) runs 256 blocks of six dependent-pair
// MFMAs with, in the gaps: nothing / four ds_read_b128 (one per gap) / + the lgkmcnt(7) waits / + two global_load_dwordx4 to AGPRs.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_block_pattern.hip -o scripts/probes/bin/mbp && scripts/probes/bin/mbp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, w, x) "v_mfma_f32_32x32x16_f16 %[" a "], " w ", %[" x "], %[" a "]\n\t"
#define RD(d, o) "ds_read_b128 %[" d "], %[ad] offset:" o "\n\t"
#define W7 "s_waitcnt lgkmcnt(7)\n\t"
#define W6 "s_waitcnt lgkmcnt(6)\n\t"
#define LD(r, o) "global_load_dwordx4 " r ", %[wv], %[wb] offset:" o "\n\t"

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(long long* out, float* sink, const char* wbase, int active) {
    __shared__ __attribute__((aligned(16))) char lds[131072 > 65536 ? 65536 : 65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    if (wave >= active) return;
    f32x16 a0, a1;
    for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
    half8 f[2][4];
    for (int b = 0; b < 2; ++b) for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) f[b][q][e] = (_Float16)(0.125f * (q + 1));
    const unsigned ad = (unsigned)(size_t)lds + lane * 16, wv = lane * 16;
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\t"
                 "v_accvgpr_write_b32 a4, %0\n\tv_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0" :: "v"(0x3c003c00) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    if (MODE >= 2)      // eight reads in flight, as at the top of a block
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:8192\n\tds_read_b128 %3, %8 offset:12288\n\t"
                     "ds_read_b128 %4, %8 offset:16384\n\tds_read_b128 %5, %8 offset:20480\n\tds_read_b128 %6, %8 offset:24576\n\tds_read_b128 %7, %8 offset:28672"
                     : "=v"(f[0][0]), "=v"(f[0][1]), "=v"(f[0][2]), "=v"(f[0][3]), "=v"(f[1][0]), "=v"(f[1][1]), "=v"(f[1][2]), "=v"(f[1][3]) : "v"(ad));
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 128; ++it) {
#define BLOCK(b, TXT) asm volatile(TXT : [a0] "+v"(a0), [a1] "+v"(a1), [l0] "+v"(f[b][0]), [l1] "+v"(f[b][1]), [h0] "+v"(f[b][2]), [h1] "+v"(f[b][3]) \
                                   : [ad] "v"(ad + ((it & 7) << 5)), [wv] "v"(wv), [wb] "s"(wbase) : "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "memory")
        if (MODE == 0) {
            BLOCK(0, MF("a0", "a[0:3]", "l0") MF("a1", "a[0:3]", "l1") MF("a0", "a[4:7]", "h0") MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") MF("a1", "a[0:3]", "h1"));
            BLOCK(1, MF("a0", "a[0:3]", "l0") MF("a1", "a[0:3]", "l1") MF("a0", "a[4:7]", "h0") MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") MF("a1", "a[0:3]", "h1"));
        } else if (MODE == 1) {     // reads in the gaps, no waits (the data is never needed in time: measures the issue cost only)
            BLOCK(0, MF("a0", "a[0:3]", "l0") RD("l0", "4096") MF("a1", "a[0:3]", "l1") RD("l1", "36864") MF("a0", "a[4:7]", "h0") MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768") "s_waitcnt lgkmcnt(0)\n\t");
            BLOCK(1, MF("a0", "a[0:3]", "l0") RD("l0", "4096") MF("a1", "a[0:3]", "l1") RD("l1", "36864") MF("a0", "a[4:7]", "h0") MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768") "s_waitcnt lgkmcnt(0)\n\t");
        } else if (MODE == 2) {     // chain3's block without the weight loads
            BLOCK(0, W7 MF("a0", "a[0:3]", "l0") RD("l0", "4096") W7 MF("a1", "a[0:3]", "l1") RD("l1", "36864") W7 MF("a0", "a[4:7]", "h0") W6 MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768"));
            BLOCK(1, W7 MF("a0", "a[0:3]", "l0") RD("l0", "4096") W7 MF("a1", "a[0:3]", "l1") RD("l1", "36864") W7 MF("a0", "a[4:7]", "h0") W6 MF("a1", "a[4:7]", "h1") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768"));
        } else {                    // chain3's block
            BLOCK(0, W7 MF("a0", "a[0:3]", "l0") RD("l0", "4096") W7 MF("a1", "a[0:3]", "l1") RD("l1", "36864") W7 MF("a0", "a[4:7]", "h0") LD("a[8:11]", "0") W6 MF("a1", "a[4:7]", "h1") LD("a[12:15]", "1024") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768"));
            BLOCK(1, W7 MF("a0", "a[0:3]", "l0") RD("l0", "4096") W7 MF("a1", "a[0:3]", "l1") RD("l1", "36864") W7 MF("a0", "a[4:7]", "h0") LD("a[8:11]", "2048") W6 MF("a1", "a[4:7]", "h1") LD("a[12:15]", "3072") MF("a0", "a[0:3]", "h0") RD("h0", "0") MF("a1", "a[0:3]", "h1") RD("h1", "32768"));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" : "+v"(a0), "+v"(a1));
    long long t1 = clock64();
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[3] + (float)f[0][0][0] + (float)f[1][3][1];
}

int main() {
    long long* d; float* s; char* w;
    (void)hipMalloc(&d, 64); (void)hipMalloc(&s, 256 * 512 * 4); (void)hipMalloc(&w, 1 << 20); (void)hipMemset(w, 0, 1 << 20);
    const char* names[4] = {"bare MFMAs", "+ 4 ds_read_b128 in gaps (wait at block end)", "+ lgkmcnt(7) waits, reads two blocks ahead", "+ 2 global_load_dwordx4 -> AGPR in gaps"};
    for (int active = 4; active <= 8; active += 4)
        for (int m = 0; m < 4; ++m) {
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) k<0><<<256, 512>>>(d, s, w, active); else if (m == 1) k<1><<<256, 512>>>(d, s, w, active);
                else if (m == 2) k<2><<<256, 512>>>(d, s, w, active); else k<3><<<256, 512>>>(d, s, w, active);
            }
            long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("%d waves/CU  %-52s cycles per MFMA: wave0 %.1f  last %.1f\n", active, names[m], h[0] / (256.0 * 6), h[active - 1] / (256.0 * 6));
        }
    return 0;
}
