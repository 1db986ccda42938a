// First-touch latency of a global load as a function of the distance to the previous touch: is the "10-13k cycles to issue eight requests at a
// new pair of tiles" of chain4's staging address translation?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/first_touch scripts/probes/first_touch.hip && /tmp/first_touch
// One wave; access i = one dword load by lane 0 at base + i * stride, waited for (s_waitcnt vmcnt(0)) and timed with s_memtime.  A fresh region of
// the buffer per stride, so every access is the first touch of its address by anyone since the buffer was written.  Also: the same with 64 lanes
// each on its own stride (one instruction, 64 translations), and a second pass over the same addresses (translations cached, data not: the
// buffer is larger than L2 + MALL and a flush pass runs in between).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ void touch_kernel(const int* base, long stride_bytes, int n, int lanes, long long* cycles, int* sink) {
    const int lane = threadIdx.x;
    int acc = 0;
    for (int i = 0; i < n; ++i) {
        const char* a = reinterpret_cast<const char*>(base) + ((long)i * lanes + (lane < lanes ? lane : 0)) * stride_bytes;
        long long t0, t1;
        int v = 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        if (lane < lanes) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        acc += v;
        if (lane == 0) cycles[i] = t1 - t0;
    }
    if (acc == 0x12345678) *sink = acc;
}

__global__ void flush_kernel(int* p, long n) { for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1; }

int main() {
    const size_t bytes = (size_t)8 << 30;
    int* buf; int* sink; long long* cyc;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4); hipMalloc(&cyc, 4096 * 8);
    hipMemset(buf, 0, bytes);
    int* fl; const long fn = (long)1 << 28; hipMalloc(&fl, fn * 4); hipMemset(fl, 0, fn * 4);
    const long strides[] = {64, 256, 1024, 4096, 16384, 65536, 262144, 2097152, 16777216};
    printf("%10s %6s | first pass: median / p90 / max cycles | second pass (after a 1 GB flush): median / p90\n", "stride", "lanes");
    size_t off = 0;
    for (int lanes : {1, 64})
        for (long st : strides) {
            int n = 256;
            while ((size_t)n * lanes * st > ((size_t)1 << 30) && n > 8) n /= 2;
            if (off + (size_t)n * lanes * st > bytes) off = 0;
            const int* base = reinterpret_cast<const int*>(reinterpret_cast<const char*>(buf) + off);
            off += ((size_t)n * lanes * st + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20) + ((size_t)64 << 20);
            std::vector<long long> h(n), h2(n);
            flush_kernel<<<1024, 256>>>(fl, fn);
            touch_kernel<<<1, 64>>>(base, st, n, lanes, cyc, sink);
            hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
            flush_kernel<<<1024, 256>>>(fl, fn);
            touch_kernel<<<1, 64>>>(base, st, n, lanes, cyc, sink);
            hipMemcpy(h2.data(), cyc, n * 8, hipMemcpyDeviceToHost);
            long long first = h[0];
            std::sort(h.begin(), h.end()); std::sort(h2.begin(), h2.end());
            printf("%10ld %6d | n=%3d first %6lld  med %6lld  p90 %6lld  max %6lld | med %6lld  p90 %6lld\n", st, lanes, n, first, h[n / 2], h[n * 9 / 10], h[n - 1], h2[n / 2], h2[n * 9 / 10]);
        }
    return 0;
}
