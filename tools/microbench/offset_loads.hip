// tools/microbench/offset_loads.hip -- what does the row search pay for its scattered start[] loads, and does the lane
// arrangement matter?  One wave iteration = 4 destinations adjacent in slot order (same pixel row, x a few pixels apart),
// each needing start[row(y + dy) + x - r] and start[row(y + dy) + x + r + 1] for dy = -7..7.
//   A  the kernel's arrangement: lane = destination * 16 + dy (lo), then the same lanes load hi   (2 instructions, 60 lanes)
//   B  row-major: lane = dy * 4 + destination (the 4 lanes of a quad read the same 64 bytes)       (2 instructions, 60 lanes)
//   C  row-major octets: lane = dy * 8 + {4 x lo, 4 x hi}                                          (2 instructions, 60 lanes)
//   D  one lane per row loads the 128-byte-aligned span as 4 x dwordx4 ... not built
//   E  no loads (address arithmetic only)
//   F  arrangement A but every lane on its own 256-byte line (the ablation of round 5)
// hipcc --offload-arch=gfx950 -O3 -o offset_loads offset_loads.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int W = 640, H = 480, B = 8, PER = 100000, R = 7;

__device__ __forceinline__ void dest_xy(int n, int &x, int &y, int &b) {
    b = n / PER;
    const int i = n - b * PER;
    y = i / 209;                 // ~209 events per pixel row
    x = (i - y * 209) * 3 + 5;
    y = min(y, H - 1);
}

template <int MODE>
__global__ __launch_bounds__(256, 7) void k(const int *__restrict__ start, int M, int *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Gd = gridDim.x, nx = 8;
    const int xcd = blockIdx.x % nx, lb = blockIdx.x / nx, bpx = Gd / nx;
    const int chunk = (M + nx - 1) / nx;
    const int per_block = (chunk + bpx - 1) / bpx;
    const int n_begin = xcd * chunk + lb * per_block;
    const int n_end = min(min(M, (xcd + 1) * chunk), n_begin + per_block);
    int acc = 0;
    int p0 = 0, p1 = 0;
    for (int n0 = n_begin + wave * 4; n0 < n_end; n0 += 16) {
        int d, dy, hi = 0;
        if (MODE == 0 || MODE == 5) { d = lane >> 4; dy = lane & 15; }
        else if (MODE == 1) { d = lane & 3; dy = lane >> 2; }
        else { d = lane & 3; hi = (lane >> 2) & 1; dy = lane >> 3; }
        int x, y, b;
        dest_xy(min(n0 + d, M - 1), x, y, b);
        int v0 = 0, v1 = 0;
        if (MODE == 2) {
            // two instructions: rows 0..7 then rows 8..14
            const int yn0 = y + dy - R, yn1 = y + dy + 8 - R;
            const int xx = hi ? min(x + R, W - 1) + 1 : max(x - R, 0);
            if (yn0 >= 0 && yn0 < H) v0 = start[W * (yn0 + H * b) + xx];
            if (dy + 8 < 15 && yn1 >= 0 && yn1 < H) v1 = start[W * (yn1 + H * b) + xx];
        } else if (MODE != 4) {
            const int yn = y + dy - R;
            if (dy < 15 && yn >= 0 && yn < H) {
                int base = W * (yn + H * b);
                if (MODE == 5) base = (lane * 64 + (n0 & 1023) * 4096) % (W * H * B - 4096);   // own 256-byte line, small region
                v0 = start[base + max(x - R, 0)];
                v1 = start[base + min(x + R, W - 1) + 1];
            }
        } else {
            const int yn = y + dy - R;
            v0 = W * (yn + H * b) + x; v1 = v0 + 15;
        }
        acc += p1 - p0;      // consume the previous iteration's loads (one iteration of latency hidden, as in the kernel)
        p0 = v0; p1 = v1;
    }
    acc += p1 - p0;
    if (acc == 0x12345678) out[0] = acc;
}

template <int MODE>
float run(const int *start, int M, int *out, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k<MODE><<<grid, 256>>>(start, M, out);
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) k<MODE><<<grid, 256>>>(start, M, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main() {
    const size_t P = (size_t)W * H * B;
    std::vector<int> h(P + 16);
    for (size_t i = 0; i < P + 16; i++) h[i] = (int)(i / 3);
    int *start, *out;
    hipMalloc(&start, (P + 16) * 4); hipMalloc(&out, 64);
    hipMemcpy(start, h.data(), (P + 16) * 4, hipMemcpyHostToDevice);
    const int M = B * PER;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int grid = prop.multiProcessorCount * 7;
    printf("grid %d blocks of 256, %d destinations\n", grid, M);
    printf("A group-major (kernel)      %.1f us\n", run<0>(start, M, out, grid));
    printf("B row-major quads           %.1f us\n", run<1>(start, M, out, grid));
    printf("C row-major octets lo|hi    %.1f us\n", run<2>(start, M, out, grid));
    printf("E no loads                  %.1f us\n", run<4>(start, M, out, grid));
    printf("F own line per lane         %.1f us\n", run<5>(start, M, out, grid));
    return 0;
}
