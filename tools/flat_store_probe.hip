// Hardware probe (gfx950): how soon after a 128-bit vector store may a VALU instruction rewrite the store's data registers?
//
// hipcc's hazard table leaves 2 wait states between {flat,global}_store_dwordx4 and a VALU write of its data registers.  Round 3 of the
// DINER kernel found rare wrong values with exactly that spacing when the store was a FLAT store and other waves of the workgroup were
// polling a word in LDS.  This probe isolates the pattern:
//
//   every wave, per iteration:  v[40:43] := pattern A, v[44:47] := pattern B
//                               STORE [p], v[40:43] ; STORE [p + 1024], v[44:47] ; s_nop NOPS ; v_mov v40, MARK     (as one asm block)
//                               ... BURST such pairs in a row (the slab stores behind a layer block)
//   then waves 0-3:             2-byte LDS stores (the geometry), arrival on a counter in LDS
//        waves 4-7:             poll that counter (ds_read_b32, s_sleep 1)                                   [POLL = 1]
//   then every wave:            wait for its stores, read them back, count the dwords that differ from the pattern
//
// for STORE in {flat_store_dwordx4, global_store_dwordx4} and NOPS in {0 (= hipcc's 2 wait states with the second store), 3}.
// Output: per variant the number of wrong dwords, how many of them carry MARK (= the overwrite came too early) and a histogram over lane & 15.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/flat_store_probe tools/flat_store_probe.hip && tools/flat_store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned MARK = 0xDEAD0000u;
constexpr int ITERS = 1024;
constexpr int BURST = 16;   // store pairs in flight before the wait (the DINER slab: 32 dwordx4 stores per wave and view)

template <bool FLAT, int NOPS, bool POLL>
__global__ __launch_bounds__(512) void probe(unsigned *buf, unsigned *wrong, unsigned *marked, unsigned *hist)
{
    __shared__ unsigned flag[64];
    __shared__ unsigned short noise[4][1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    unsigned *base = buf + ((size_t)(blockIdx.x * 8 + wave) * 64) * 512 + lane * 4;
    unsigned nwrong = 0, nmark = 0;
    for (int it = 0; it < ITERS; ++it) {
        __syncthreads();   // every wave starts its burst together (like the end of a layer block)
#pragma unroll
        for (int k = 0; k < BURST; ++k) {
            unsigned *p = base + ((it * BURST + k) & 63) * 512;
            const unsigned a = ((unsigned)(it * BURST + k) << 8) | 0x10u, b = ((unsigned)(it * BURST + k) << 8) | 0x20u;
            if (FLAT)
                asm volatile("v_add_u32 v40, %1, 0\n\tv_add_u32 v41, %1, 1\n\tv_add_u32 v42, %1, 2\n\tv_add_u32 v43, %1, 3\n\t"
                             "v_add_u32 v44, %2, 0\n\tv_add_u32 v45, %2, 1\n\tv_add_u32 v46, %2, 2\n\tv_add_u32 v47, %2, 3\n\t"
                             "s_nop 4\n\t"
                             "flat_store_dwordx4 %0, v[40:43]\n\t"
                             "flat_store_dwordx4 %0, v[44:47] offset:1024\n\t"
                             "s_nop %c4\n\t"
                             "v_mov_b32 v40, %3\n\t"
                             "v_mov_b32 v44, %3\n\t"
                             :: "v"(p), "v"(a), "v"(b), "v"(MARK), "n"(NOPS) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            else
                asm volatile("v_add_u32 v40, %1, 0\n\tv_add_u32 v41, %1, 1\n\tv_add_u32 v42, %1, 2\n\tv_add_u32 v43, %1, 3\n\t"
                             "v_add_u32 v44, %2, 0\n\tv_add_u32 v45, %2, 1\n\tv_add_u32 v46, %2, 2\n\tv_add_u32 v47, %2, 3\n\t"
                             "s_nop 4\n\t"
                             "global_store_dwordx4 %0, v[40:43], off\n\t"
                             "global_store_dwordx4 %0, v[44:47], off offset:1024\n\t"
                             "s_nop %c4\n\t"
                             "v_mov_b32 v40, %3\n\t"
                             "v_mov_b32 v44, %3\n\t"
                             :: "v"(p), "v"(a), "v"(b), "v"(MARK), "n"(NOPS) : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        }
        if (wave >= 4) {
            if (POLL) {   // the polling partner: waits for waves 0-3 of this iteration (arrival counter in LDS, ds_read_b32 + s_sleep 1)
                unsigned addr = (unsigned)(uintptr_t)flag, v, sg, spin;
                const unsigned tgt = 4u * (unsigned)(it + 1);
                asm volatile("s_mov_b32 %2, 4000000\n"
                             "W_%=:\n\tds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 %1, %0\n\ts_cmp_ge_u32 %1, %4\n\ts_cbranch_scc1 D_%=\n\t"
                             "s_sub_u32 %2, %2, 1\n\ts_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 D_%=\n\ts_sleep 1\n\ts_branch W_%=\nD_%=:"
                             : "=&v"(v), "=&s"(sg), "=&s"(spin) : "v"(addr), "s"(tgt) : "memory", "scc");
            }
        } else {   // the geometry waves: 2-byte LDS stores, then the arrival
#pragma unroll 1
            for (int e = 0; e < 16; ++e) {
                noise[wave][lane * 8 + (e & 7)] = (unsigned short)(it + e);
                noise[wave][512 + lane * 8 + (e & 7)] = (unsigned short)(it - e);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(flag, 1u);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (int k = 0; k < BURST; ++k) {
            unsigned *p = base + ((it * BURST + k) & 63) * 512;
            const unsigned a = ((unsigned)(it * BURST + k) << 8) | 0x10u, b = ((unsigned)(it * BURST + k) << 8) | 0x20u;
            const u32x4 ra = __builtin_nontemporal_load((const u32x4 *)p), rb = __builtin_nontemporal_load((const u32x4 *)(p + 256));
            const unsigned got[8] = {ra[0], ra[1], ra[2], ra[3], rb[0], rb[1], rb[2], rb[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned want = (j < 4 ? a : b) + (j & 3);
                if (got[j] != want) {
                    ++nwrong;
                    if (got[j] == MARK) ++nmark;
                }
            }
        }
    }
    if (nwrong) {
        atomicAdd(wrong, nwrong);
        atomicAdd(marked, nmark);
        atomicAdd(hist + (lane & 15), nwrong);
    }
}

template <bool FLAT, int NOPS, bool POLL> static void run(const char *name, unsigned *buf, unsigned *ctr)
{
    (void)hipMemset(ctr, 0, 64 * sizeof(unsigned));
    hipLaunchKernelGGL((probe<FLAT, NOPS, POLL>), dim3(256), dim3(512), 0, 0, buf, ctr, ctr + 1, ctr + 16);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    unsigned h[64];
    (void)hipMemcpy(h, ctr, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s wrong dwords %8u  (MARK %8u) of %.3g  | by lane&15:", name, h[0], h[1], 256.0 * 8 * 64 * ITERS * BURST * 8);
    for (int i = 0; i < 16; ++i) printf(" %u", h[16 + i]);
    printf("\n");
}

int main()
{
    unsigned *buf, *ctr;
    if (hipMalloc(&buf, (size_t)256 * 8 * 64 * 512 * sizeof(unsigned)) != hipSuccess || hipMalloc(&ctr, 64 * sizeof(unsigned)) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        run<true, 0, true>("flat_store,   2 wait states, partner polls", buf, ctr);
        run<true, 0, false>("flat_store,   2 wait states, partner idle", buf, ctr);
        run<true, 3, true>("flat_store,   5 wait states, partner polls", buf, ctr);
        run<false, 0, true>("global_store, 2 wait states, partner polls", buf, ctr);
        run<false, 0, false>("global_store, 2 wait states, partner idle", buf, ctr);
        run<false, 3, true>("global_store, 5 wait states, partner polls", buf, ctr);
    }
    return 0;
}
