// Read-bandwidth probe for the MI355X box (standalone; build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o tools/hbm_probe).
// Answers three questions the step kernels are designed around:
//   1. what does a well-formed read-only stream reach from HBM (vector loads vs LDS-DMA, loads in flight, grid shape)?
//   2. what can ONE compute unit ingest (few work-groups, HBM- or L2-resident source)?
//   3. what does the Infinity Cache serve (128 MB window re-read)?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid-stride 16-byte loads, U independent loads per thread in flight
template <int U>
__global__ void read_vec(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// work-group owns a contiguous chunk (like a GEMM weight slab); each wave streams 1-KiB pieces into an LDS ring by
// LDS-DMA with DEPTH pieces in flight; LDS is never read back (pure ingest)
template <int DEPTH>
__global__ void read_dma(const unsigned char* __restrict__ p, size_t bytes_per_wg, uint32_t* sink, size_t wg_stride = (size_t)-1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = p + (size_t)blockIdx.x * (wg_stride == (size_t)-1 ? bytes_per_wg : wg_stride);
    const size_t pieces = bytes_per_wg / 1024;
    unsigned char* ring = smem + wave * DEPTH * 1024;
    size_t pc = wave;
    int slot = 0;
    for (; pc < pieces; pc += nw) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + pc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, 0);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        if (slot == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[lane] == 0x7f && ring[lane + 64] == 0x11 && lane == 99) sink[0] = 1;
}


// GEMM-like access: W[N][K] row-major (ld bytes per row); work-group (n-block, k-slice) reads BN rows x CH contiguous
// bytes per step, walking along K.  CH = 128 is the skinny GEMM's BK = 64 bf16 pattern.
template <int DEPTH>
__global__ void read_strided(const unsigned char* __restrict__ p, int bn, int ch, size_t ld, size_t slice_bytes, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* base = p + (size_t)blockIdx.x * bn * ld + (size_t)blockIdx.y * slice_bytes;
    unsigned char* ring = smem + wave * DEPTH * 1024;
    const int lanes_per_row = ch / 16, rows_per_piece = 64 / lanes_per_row;
    const int pieces_per_step = bn / rows_per_piece;
    const int steps = (int)(slice_bytes / ch);
    int slot = 0;
    for (int st = 0; st < steps; ++st)
        for (int pc = wave; pc < pieces_per_step; pc += nw) {
            const int row = pc * rows_per_piece + lane / lanes_per_row;
            const unsigned char* src = base + (size_t)row * ld + (size_t)st * ch + (lane % lanes_per_row) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, 0);
            slot = slot + 1 == DEPTH ? 0 : slot + 1;
            if (slot == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ring[lane] == 0x7f && ring[lane + 64] == 0x11 && lane == 99) sink[0] = 1;
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

int main(int argc, char** argv) {
    const size_t total = (size_t)4 << 30;                 // 4 GiB pool
    unsigned char* buf;
    uint32_t* sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, total));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, cus);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto run = [&](const char* name, size_t window, size_t span, int reps, auto launch) {
        // window = bytes read per launch; launches rotate through `span` bytes of the pool (span == window: cache-resident re-read)
        for (int w = 0; w < 2; ++w) launch(buf);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        size_t off = 0;
        for (int r = 0; r < reps; ++r) {
            launch(buf + off);
            off += window;
            if (off + window > span) off = 0;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        const float ms = time_ms(e0, e1) / reps;
        printf("%-64s %8.1f us  %8.1f GB/s\n", name, ms * 1e3, window / ms / 1e6);
        fflush(stdout);
    };

    char nm[160];
    CK(hipFuncSetAttribute((const void*)read_dma<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)read_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // ---- 1. HBM stream: 512 MiB windows rotating through 4 GiB ----
    const size_t W = (size_t)512 << 20;
    for (int block : {256, 512, 1024})
        for (int wgs_per_cu : {1, 2, 4, 8}) {
            if ((size_t)block * wgs_per_cu > 2048) continue;
            const int grid = cus * wgs_per_cu;
            snprintf(nm, sizeof nm, "hbm vec U=4  block=%4d grid=%5d", block, grid);
            run(nm, W, total, 8, [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<4>, dim3(grid), dim3(block), 0, 0, (const u32x4*)p, W / 16, sink); });
            snprintf(nm, sizeof nm, "hbm vec U=8  block=%4d grid=%5d", block, grid);
            run(nm, W, total, 8, [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<8>, dim3(grid), dim3(block), 0, 0, (const u32x4*)p, W / 16, sink); });
        }
    {
        const int block = 256;
        const int grid = (int)(W / 16 / block / 8);
        snprintf(nm, sizeof nm, "hbm vec U=8  block=%4d grid=%5d (one pass per thread)", block, grid);
        run(nm, W, total, 8, [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<8>, dim3(grid), dim3(block), 0, 0, (const u32x4*)p, W / 16, sink); });
    }
    for (int block : {256, 512})
        for (int grid_mul : {1, 2, 4, 16}) {
            const int grid = cus * grid_mul;
            const size_t per_wg = W / grid / 1024 * 1024;
            snprintf(nm, sizeof nm, "hbm dma depth=8  block=%4d grid=%5d", block, grid);
            run(nm, per_wg * grid, total, 8, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<8>, dim3(grid), dim3(block), (block / 64) * 8 * 1024, 0, p, per_wg, sink); });
            snprintf(nm, sizeof nm, "hbm dma depth=16 block=%4d grid=%5d", block, grid);
            run(nm, per_wg * grid, total, 8, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(block), (block / 64) * 16 * 1024, 0, p, per_wg, sink); });
        }
    // ---- 1b. GEMM-sized streams: 100 MB / 34 MB (one projection), rotating ----
    for (size_t mb : {34, 100, 180}) {
        const size_t win = mb << 20;
        for (int grid_mul : {1, 2, 4}) {
            const int grid = cus * grid_mul, block = 512;
            const size_t per_wg = win / grid / 1024 * 1024;
            snprintf(nm, sizeof nm, "hbm dma depth=16 block= 512 grid=%5d  window=%zu MB", grid, mb);
            run(nm, per_wg * grid, total, 40, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(block), (block / 64) * 16 * 1024, 0, p, per_wg, sink); });
            snprintf(nm, sizeof nm, "hbm vec U=8     block= 512 grid=%5d  window=%zu MB", grid, mb);
            run(nm, win, total, 40, [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<8>, dim3(grid), dim3(block), 0, 0, (const u32x4*)p, win / 16, sink); });
        }
    }
    // ---- 1c. GEMM-like strided access over a [12288][4096] bf16 matrix (100 MB), rotating through 32 of them ----
    CK(hipFuncSetAttribute((const void*)read_strided<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {
        const size_t N = 12288, ld = 8192, mat = N * ld;
        for (int bn : {128, 256})
            for (int ch : {128, 256, 512, 1024})
                for (int S : {2, 4, 5, 8}) {
                    const int nblk = (int)(N / bn);
                    const size_t slice = ld / S / ch * ch;
                    snprintf(nm, sizeof nm, "strided W[12288][4096]: BN=%3d bytes/row/step=%4d split=%d grid=%d", bn, ch, S, nblk * S);
                    run(nm, (size_t)nblk * bn * slice * S, mat * 32, 40,
                        [&](unsigned char* p) { hipLaunchKernelGGL(read_strided<8>, dim3(nblk, S), dim3(512), 8 * 8 * 1024, 0, p, bn, ch, ld, slice, sink); });
                }
    }
    // ---- 2. per-CU ingest: 8 / 32 work-groups only ----
    for (int grid : {8, 32}) {
        const size_t win = (size_t)grid << 22;            // 4 MiB per work-group
        snprintf(nm, sizeof nm, "few WGs, HBM source: dma depth=16 block=512 grid=%d (per WG = /%d)", grid, grid);
        run(nm, win, total, 50, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(512), 8 * 16 * 1024, 0, p, (size_t)1 << 22, sink); });
        snprintf(nm, sizeof nm, "few WGs, HBM source: vec U=8 block=512 grid=%d", grid);
        run(nm, win, total, 50, [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<8>, dim3(grid), dim3(512), 0, 0, (const u32x4*)p, win / 16, sink); });
        snprintf(nm, sizeof nm, "few WGs, cache-resident source: dma depth=16 grid=%d", grid);
        run(nm, win, win, 50, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(512), 8 * 16 * 1024, 0, p, (size_t)1 << 22, sink); });
    }
    // ---- 2b. all CUs re-reading the same small block (an activation tile shared by every work-group) ----
    for (size_t kb : {480, 2048}) {
        const int grid = cus * 2;
        const size_t per = kb * 1024;
        snprintf(nm, sizeof nm, "all %d WGs read the SAME %zu KiB block (aggregate ingest)", grid, kb);
        run(nm, per * grid, per * grid, 50, [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(512), 8 * 16 * 1024, 0, buf, per, sink, (size_t)0); });
    }
    // ---- 3. Infinity Cache: 128 MiB window re-read in place ----
    {
        const size_t win = (size_t)128 << 20;
        const int grid = cus * 4, block = 512;
        const size_t per_wg = win / grid / 1024 * 1024;
        run("infinity-cache resident 128 MiB re-read: dma depth=16", per_wg * grid, per_wg * grid, 40,
            [&](unsigned char* p) { hipLaunchKernelGGL(read_dma<16>, dim3(grid), dim3(block), (block / 64) * 16 * 1024, 0, p, per_wg, sink); });
        run("infinity-cache resident 128 MiB re-read: vec U=8", win, win, 40,
            [&](unsigned char* p) { hipLaunchKernelGGL(read_vec<8>, dim3(grid), dim3(block), 0, 0, (const u32x4*)p, win / 16, sink); });
    }
    return 0;
}
