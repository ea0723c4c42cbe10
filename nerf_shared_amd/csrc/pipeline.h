// pipeline.h -- the weight-stream pipeline shared by the fused bf16 MLP kernels:
// LDS-DMA ring, counted-vmcnt block syncs, read-ahead queue (see mlp_bf16.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>

namespace na {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// Compile-time shape of the weight pipeline.
//   WAVES  waves per workgroup (each owns 32 points)
//   BF     fragments per ring block (one workgroup barrier per block)
//   NS     ring slots (blocks resident in LDS)
//   PHASE  0: the sync for block b sits at its first fragment (drains this wave's LDS reads).
//          p>0: the sync that publishes block b+1 sits p fragments into block b, so the
//          reads of the next block's first fragments are not fenced behind a barrier.
//   LA     with PHASE > 0: A fragments are read LA MFMAs ahead of their use through a register
//          queue, so the reads for the MFMAs right after a barrier are already in flight
//          when the barrier is reached (legal while LA <= PHASE <= BF - LA).
//   ABL    timing-only ablations for A/B measurements (results are WRONG when non-zero):
//          1 = no syncs/DMA, 2 = no LDS fragment reads, 4 = no positional encoding
//   NP     32-point column tiles per wave: every A fragment read from LDS feeds NP MFMAs
//          (NP = 2 halves the LDS read traffic; needs one wave per SIMD for its registers)
//   OPT    code-generation options for A/B runs: 1 = ReLU on fp32 registers (one integer max per
//          value) instead of one packed 16-bit max per converted pair
//          2 = STAGGER: waves WAVES/2.. (the SIMD partners of waves 0..WAVES/2-1) take every block sync half
//          a block later in their program -- at fragment 0 of block b+1 instead of fragment PHASE of block b --
//          so the two waves of a SIMD run half a block apart instead of in lockstep: one converts / reads
//          LDS / parks at the barrier while the other has the matrix pipe (MI355X_MICROARCH.md, "Two waves per
//          SIMD", item 9).  Same barrier count, same DMA schedule; legal because the late half, at
//          fragment 0 of block b+1, has finished every read of block b-1... see maybe_sync.
//   LEDGER a compile-time count of the compiler-issued global stores that sit in program order
//          before each fragment (training kernels).  Those stores share the in-order vmcnt queue
//          with the ring DMA; the counted waits add the stores known to be younger than the block
//          they wait for, so the younger DMA blocks stay in flight instead of being drained.
//          A ledger may under-count (the wait only gets stricter) but must never over-count.
struct NoLedger {
    static constexpr int stores_before(int) { return 0; }
};
template <int WAVES_, int BF_, int NS_, int PHASE_, int LA_ = 0, int ABL_ = 0, int NP_ = 1, int OPT_ = 0, class LEDGER_ = NoLedger>
struct Ctx {
    using Ledger = LEDGER_;
    static constexpr int WAVES = WAVES_, BF = BF_, NS = NS_, PHASE = PHASE_, LA = LA_, ABL = ABL_, NP = NP_, OPT = OPT_;
    static constexpr int WAVES_PER_SIMD = (WAVES_ * NP_ >= 8 && NP_ == 1) ? 2 : 1;
    static_assert(LA_ == 0 || (PHASE_ > 0 && LA_ <= PHASE_ && PHASE_ + LA_ <= BF_), "read-ahead would cross an unpublished block");
    bf16x8 q[LA_ > 0 ? LA_ : 1];
    static constexpr int PIECES = BF / WAVES;            // 1-KiB DMA pieces per wave per block
    static constexpr int BLOCK_BYTES = BF * 1024;
    static constexpr int RING_BYTES = NS * BLOCK_BYTES;
    static constexpr int LOOKAHEAD = PHASE > 0 ? 1 : 0;  // a sync at block b publishes block b + LOOKAHEAD
    static constexpr int N_PHASES = (OPT_ & 128) ? 4 : (OPT_ & 8) ? 2 : 1;   // SPLIT_DMA issue phases (late_issue)
    static_assert(BF % WAVES == 0 && NS >= 3 && PHASE < BF, "bad pipeline shape");
    const char *gstream;     // this lane's view of the fragment stream (base + lane*16)
    const char *ring_lane;   // LDS ring + lane*16
    uint32_t ring_u32;       // LDS byte address of the ring
    const float *bias_half;  // LDS bias table + (lane>>5)*16
    int wave;
    int lag;                 // STAGGER: 1 for waves WAVES/2.. (wave-uniform)
    int phase;               // SPLIT_DMA: this wave's issue phase 0..3 (wave-uniform; SIMD partners w, w+4 differ by 2)
    // OPT & 32 (CONTINUOUS): the weight stream does not stop at a tile boundary.  The blocks of consecutive tiles
    // are numbered through (V = t NB + b, ring slot V mod NS), so the first blocks of the next tile are fetched
    // under the last blocks of the current one and nothing drains in between.  Inside a tile block b is reached
    // through slot_*[b mod NS]; the kernel rotates the table by NB mod NS slots when it moves to its next tile.
    uint32_t slot_lane[NS_];   // LDS byte address of a slot + lane*16 (fragment reads)
    uint32_t slot_u32[NS_];    // LDS byte address of a slot (DMA destination, wave-uniform)
    const char *gstream_next;  // the next tile's stream + lane*16 (the same model here)
    int has_next;              // this workgroup has another tile after the current one (workgroup-uniform)
    // OPT & 64 (BUFFER_DMA): the ring DMA as buffer_load ... lds: a buffer resource over the stream (scalar), one constant
    // VGPR (lane * 16) and a scalar offset per block instead of a 64-bit per-lane address pair and two VALU per piece;
    // the instruction's immediate offset advances the memory AND the LDS address, so a wave's pieces of a block are
    // consecutive immediates.  Out-of-range reads return zero instead of faulting (num_records = stream bytes).
    typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
    rsrc_t rsrc, rsrc_next;    // buffer resources of this tile's / the next tile's stream
    unsigned lane16;           // lane * 16
    unsigned wave_off;         // wave * PIECES * 1024: this wave's first piece inside a block (memory and LDS)
};

// Raw buffer resource over `bytes` bytes at `base` (gfx9 dword 3: 32-bit data format, no swizzle).
__device__ __forceinline__ __attribute__((ext_vector_type(4))) unsigned make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)(uintptr_t)base;
    __attribute__((ext_vector_type(4))) unsigned r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

typedef __attribute__((address_space(3))) const bf16x8 lds_frag_t;
typedef __attribute__((address_space(3))) volatile uint32_t lds_u32_t;      // an LDS word addressed as LDS (no generic-pointer cast)

// LDS-DMA of this wave's share of stream block BB into its ring slot.  SRC >= 0: block SRC of the NEXT tile's
// stream goes into the slot that block BB of the through-numbered stream owns (CONTINUOUS).
// SEL: 0 = every wave issues; 1 + p = only the waves whose c.phase == p (SPLIT_DMA: p = 0 right behind the sync,
// p = 2 half a block later; with four phases p = 1, 3 a quarter block in between).  The selection is a
// scalar branch INSIDE the asm statement: to the compiler the call site stays straight-line code (a C++ `if` around the
// DMA splits the kernel's one large scheduling region and costs the view-branch kernels ~50 VGPRs and spills).
template <int BB, class C, int SRC = -1, int SEL = 0>
__device__ __forceinline__ void issue_block(const C &c) {
    constexpr int slot = BB % C::NS;
    if constexpr ((C::OPT & 64) != 0) {
        static_assert(C::PIECES * 1024 <= 4096, "a wave's pieces must fit the 12-bit immediate offset");
        const unsigned soff = __builtin_amdgcn_readfirstlane(c.wave_off + (unsigned)(SRC >= 0 ? SRC : BB) * C::BLOCK_BYTES);
        const unsigned l0 = __builtin_amdgcn_readfirstlane(((C::OPT & 32) != 0 ? c.slot_u32[slot] : c.ring_u32 + slot * C::BLOCK_BYTES) + c.wave_off);
        const int lag_s = SEL == 0 ? 0 : __builtin_amdgcn_readfirstlane(c.phase);
        const auto rs = SRC >= 0 ? c.rsrc_next : c.rsrc;
        unsigned keep;
        // one asm statement per block: M0 is set once, the pieces are consecutive immediates
#define NA_BDMA2 "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t" "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
#define NA_BDMA4 NA_BDMA2 "buffer_load_dwordx4 %1, %2, %4 offen offset:2048 lds\n\t" "buffer_load_dwordx4 %1, %2, %4 offen offset:3072 lds\n\t"
#define NA_BDMA_ASM(LOADS)                                                                                               \
    do {                                                                                                                  \
        if constexpr (SEL == 0) {                                                                                         \
            asm volatile("s_mov_b32 %0, m0\n\t" "s_mov_b32 m0, %3\n\t" "s_nop 0\n\t" LOADS "s_mov_b32 m0, %0"            \
                         : "=&s"(keep) : "v"(c.lane16), "s"(rs), "s"(l0), "s"(soff) : "memory");                           \
        } else {                                                                                                          \
            asm volatile("s_cmp_lg_u32 %5, %6\n\t" "s_cbranch_scc1 .Lskip_bdma_%=\n\t" "s_mov_b32 %0, m0\n\t"              \
                         "s_mov_b32 m0, %3\n\t" "s_nop 0\n\t" LOADS "s_mov_b32 m0, %0\n" ".Lskip_bdma_%=:"                 \
                         : "=&s"(keep) : "v"(c.lane16), "s"(rs), "s"(l0), "s"(soff), "s"(lag_s), "n"(SEL - 1)              \
                         : "memory", "scc");                                                                              \
        }                                                                                                                 \
    } while (0)
        static_assert(C::PIECES == 2 || C::PIECES == 4, "pieces per wave and block");
        if constexpr (C::PIECES == 2) NA_BDMA_ASM(NA_BDMA2);
        else NA_BDMA_ASM(NA_BDMA4);
#undef NA_BDMA_ASM
        return;
    }
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
        const int piece = c.wave * C::PIECES + i;
        const char *g = (SRC >= 0 ? c.gstream_next + (size_t)SRC * C::BLOCK_BYTES : c.gstream + (size_t)BB * C::BLOCK_BYTES) + piece * 1024;
        const uint32_t l = ((C::OPT & 32) != 0 ? c.slot_u32[slot] : c.ring_u32 + slot * C::BLOCK_BYTES) + piece * 1024;   // wave-uniform
        unsigned keep;
        const int lag_s = SEL == 0 ? 0 : __builtin_amdgcn_readfirstlane(c.phase);   // certainly an SGPR for the asm's s_cmp
        if constexpr (SEL == 0) {
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(g), "s"(l)
                : "memory");
        } else {
            asm volatile(
                "s_cmp_lg_u32 %3, %4\n\t"
                "s_cbranch_scc1 .Lskip_dma_%=\n\t"
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "s_mov_b32 m0, %0\n"
                ".Lskip_dma_%=:"
                : "=&s"(keep)
                : "v"(g), "s"(l), "s"(lag_s), "n"(SEL - 1)
                : "memory", "scc");
        }
    }
}

// The DMA that sync S_B starts: block B+NS-1 of this tile, or (CONTINUOUS, with a next tile) the block of the next tile
// that owns the same ring slot.  PH: the issue phase of this call site (0 = right behind the barrier).  Without
// SPLIT_DMA every wave issues at phase 0; with it only the waves of that phase do.
template <int B, int NB, int PH, class C>
__device__ __forceinline__ void sync_issue(const C &c) {
    constexpr int BB = B + C::NS - 1;
    constexpr bool split = C::N_PHASES > 1;
    if constexpr (PH != 0 && !split) return;
    constexpr int SEL = !split ? 0 : 1 + PH;
    if constexpr (BB < NB) {
        issue_block<BB, C, -1, SEL>(c);
    } else if constexpr ((C::OPT & 32) != 0 && BB - NB < C::NS - 1 - C::LOOKAHEAD) {
        // the next tile's blocks 0 .. NS-2-LOOKAHEAD are what its own prologue would have issued
        if (c.has_next) issue_block<BB, C, BB - NB, SEL>(c);
    }
}

// Sync S_b (b >= -LOOKAHEAD): this wave's pieces of block b+LOOKAHEAD have landed
// (counted vmcnt: younger blocks stay in flight), everyone agrees (s_barrier), then
// the slot of block b-1 -- which every wave has finished reading -- is refilled with
// block b+NS-1.  With PHASE == 0 the slot being refilled was read up to the previous
// instruction, so this wave's LDS reads are drained first (lgkmcnt(0)); with PHASE > 0
// its last read is PHASE MFMAs old and already consumed, and only instruction motion
// across the sync has to be prevented.
template <int B, int NB, class C>
__device__ __forceinline__ void block_sync(const C &c) {
    if constexpr (C::ABL & 1) return;
    constexpr int need = B + C::LOOKAHEAD;
    constexpr int last_issued = (B + C::NS - 2) < (NB - 1) ? (B + C::NS - 2) : (NB - 1);
    // stores issued since the sync that started block `need`'s DMA (sync -1 sits before fragment 0)
    constexpr int issue_sync = need - (C::NS - 1);
    constexpr int pos_now = B >= 0 ? B * C::BF + C::PHASE : 0;
    constexpr int pos_then = issue_sync >= 0 ? issue_sync * C::BF + C::PHASE : 0;
    constexpr int younger_stores = (C::PHASE > 0 && issue_sync >= -1 && last_issued > need)
                                       ? C::Ledger::stores_before(pos_now) - C::Ledger::stores_before(pos_then) : 0;
    static_assert(younger_stores >= 0, "ledger must be monotonic");
    constexpr int cnt_raw = (last_issued > need ? last_issued - need : 0) * C::PIECES + younger_stores;
    constexpr int cnt = cnt_raw <= 63 ? cnt_raw : 63;    // vmcnt is 6 bits; a smaller count only waits for more
    // CONTINUOUS: blocks NB, NB+1, ... are the next tile's first blocks, in flight only if there is a next tile.  A sync
    // whose allowance would include one of them (B + NS - 2 >= NB) picks its count at run time.
    constexpr bool next_in_window = (C::OPT & 32) != 0 && C::PHASE > 0 && (B + C::NS - 2) >= NB;
    constexpr int cnt_next = ((B + C::NS - 2) - need) * C::PIECES;
    static_assert(cnt_next <= 63, "vmcnt field is 6 bits");
    if constexpr (C::PHASE == 0) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(cnt) : "memory");
    } else if constexpr (next_in_window) {
        __builtin_amdgcn_sched_barrier(0);
        if (c.has_next) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(cnt_next) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(cnt) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    } else {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(cnt) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    sync_issue<B, NB, 0>(c);
}

// SPLIT_DMA (OPT & 8: two phases, OPT & 128: four): the waves of phase p issue their DMA pieces of block b+NS-1 p quarter
// blocks after sync S_b instead of right behind its barrier (phases 0 and 2 with two phases), so the two waves of a SIMD
// -- whose phases differ by 2 -- are never both inside the slow LDS-DMA issue path, and with four phases only two of the
// CU's eight waves are at any time.  The pieces are still issued between S_b and S_b+1 in every wave's program order, so
// the counted vmcnt waits are unchanged.
template <int n, int NB, int NFRAGS, class C>
__device__ __forceinline__ void late_issue(const C &c) {
    if constexpr (C::N_PHASES > 1 && C::PHASE > 0) {
        static_for<3>([&](auto i_) {
            constexpr int ph = i_ + 1;                                  // phases 1, 2, 3
            if constexpr (C::N_PHASES == 4 || ph == 2) {
                constexpr int off = C::PHASE + ph * (C::BF / 4);            // fragments behind the block start of the sync's block
                if constexpr ((n - off) % C::BF == 0 && n - off >= -C::BF) {
                    constexpr int B = (n - off) / C::BF;                    // the sync this issue belongs to (n = off - BF: the prologue's sync -1)
                    if constexpr (B >= -1 && B + C::LOOKAHEAD < NB) sync_issue<B, NB, ph>(c);
                }
                // a phase whose slot behind the prologue's sync -1 would lie in front of fragment 0 issues at fragment 0
                if constexpr (n == 0 && off < C::BF && -1 + C::LOOKAHEAD < NB) sync_issue<-1, NB, ph>(c);
                // ... and one whose slot behind the tile's last syncs would lie past the last fragment issues at the last one
                if constexpr (n == NFRAGS - 1) {
                    static_for<2>([&](auto j_) {
                        constexpr int B = NB - 2 - j_;
                        if constexpr (B >= 0 && B * C::BF + off > NFRAGS - 1) sync_issue<B, NB, ph>(c);
                    });
                }
            }
        });
    }
}

template <int NB, class C>
__device__ __forceinline__ void pipeline_prologue(const C &c) {
    if constexpr (C::ABL & 1) return;
    static_for<C::NS - 1 - C::LOOKAHEAD>([&](auto b_) { constexpr int b = b_; if constexpr (b < NB) issue_block<b>(c); });
}

template <int n, class C>
__device__ __forceinline__ bf16x8 ring_frag(const C &c) {
    if constexpr ((C::OPT & 32) != 0)
        return *(lds_frag_t *)(uintptr_t)(c.slot_lane[(n / C::BF) % C::NS] + ((n % C::BF) << 10));
    else
        return *reinterpret_cast<const bf16x8 *>(c.ring_lane + ((n % (C::NS * C::BF)) << 10));
}

// Sync S_b publishes block b+1 and refills the slot of block b-1 with block b+NS-1.  The early waves reach it
// PHASE fragments into block b.  With STAGGER the late waves reach the same barrier at fragment PHASE of block
// b too -- in wall time -- but their program is half a block behind: for them it sits at fragment
// PHASE - BF/2 of block b.  Legal: a late wave at that point has consumed every fragment of block b-1 (reads
// run LA <= BF/2 - ... fragments ahead and are waited for before their MFMA), and it needs block b+1 only
// BF/2 fragments later than the early waves do.
template <int n, int NB, class C>
__device__ __forceinline__ void maybe_sync(const C &c) {
    if constexpr ((C::OPT & 2) != 0) {
        static_assert(C::PHASE == C::BF / 2 && C::LA <= C::PHASE, "stagger is written for a mid-block sync");
        if constexpr (n % C::BF == C::PHASE && (n / C::BF + 1) < NB) {
            if (!c.lag) block_sync<n / C::BF, NB>(c);
        }
        if constexpr (n % C::BF == 0 && (n / C::BF + 1) < NB) {
            if (c.lag) block_sync<n / C::BF, NB>(c);
        }
    } else {
        if constexpr (n % C::BF == C::PHASE && (n / C::BF + C::LOOKAHEAD) < NB) block_sync<n / C::BF, NB>(c);
    }
}

// Fragment n of the stream, in consumption order (syncs included).
template <int n, int NB, int NFRAGS, class C>
__device__ __forceinline__ bf16x8 take(C &c) {
    maybe_sync<n, NB>(c);
    late_issue<n, NB, NFRAGS>(c);
    if constexpr (C::ABL & 2) {
        bf16x8 f = c.q[0];
        asm volatile("" : "+v"(f));     // opaque: keeps one MFMA per fragment without an LDS read
        return f;
    } else if constexpr (C::LA == 0) {
        return ring_frag<n>(c);
    } else {
        const bf16x8 f = c.q[n % C::LA];
        if constexpr (n + C::LA < NFRAGS) c.q[n % C::LA] = ring_frag<n + C::LA>(c);
        return f;
    }
}

// Fragments [F0, F0 + N) of the stream pass unread (a product the caller does not need -- the dX chains' encoding products
// when no ray gradient is asked for): their syncs and DMA issue sites stay where take<n> has them, and the read-ahead queue
// is refilled so that fragment F0 + N is the next one out.
template <int F0, int N, int NB, int NFRAGS, class C>
__device__ __forceinline__ void skip_frags(C &c) {
    static_for<N>([&](auto i_) {
        constexpr int n = F0 + i_;
        maybe_sync<n, NB>(c);
        late_issue<n, NB, NFRAGS>(c);
        if constexpr (C::LA > 0 && (C::ABL & 2) == 0) {
            if constexpr (n + C::LA >= F0 + N && n + C::LA < NFRAGS) c.q[n % C::LA] = ring_frag<n + C::LA>(c);
        }
    });
}

// o + d * z with the product rounded before the sum, as the reference's two ATen ops do (render_utils.py:131).  HIP's
// __fmul_rn / __fadd_rn are plain operators to the compiler, which contracts them into one v_fma (single rounding) under
// the default -ffp-contract=fast -- the round-1 kernels did, in the rays + depths mode.  The empty asm makes the product
// opaque, so it stays a product.
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
    float m = a * b;
    asm volatile("" : "+v"(m));
    return m + c;
}

// ReLU as one integer max on the fp32 bits (negative floats are negative ints);
// fmaxf would cost a second v_max to canonicalise a possible sNaN.
__device__ __forceinline__ float relu_bits(float v) {
    int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

}  // namespace na
