#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = 0;
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    for (int threads : {192, 256, 320}) {
        const int nb = 512;
        unsigned* d; hipMalloc(&d, nb * 8 * 2 * 4); hipMemset(d, 0xff, nb * 8 * 2 * 4);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 60000);
        probe<<<nb, threads, 55000>>>(d, 2000);   // 20 us spin: all blocks co-resident
        hipDeviceSynchronize();
        std::vector<unsigned> h(nb * 8 * 2); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("threads %d\n", threads);
        std::map<unsigned, std::vector<int>> cu_blocks;
        for (int b = 0; b < nb; ++b) {
            unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xf;
            unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            cu_blocks[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        }
        printf("  distinct CUs used: %zu\n", cu_blocks.size());
        int shown = 0;
        for (auto& kv : cu_blocks) {
            if (shown++ >= 4) break;
            printf("  cu %04x blocks:", kv.first);
            for (int b : kv.second) {
                printf(" %d[simd", b);
                for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[(b * 8 + w) * 2] >> 4) & 3);
                printf("]");
            }
            printf("\n");
        }
        for (int b : {0, 1, 2, 8, 9, 256, 257}) printf("  block %d: xcc %u hw %08x\n", b, h[b * 16 + 1] & 0xf, h[b * 16]);
        hipFree(d);
    }
    return 0;
}
