// The align stage as a sequence of specialised kernels with on-chip state (the common case of aligner::align after seeding,
// include/aligner/aligner_ksw2.hpp:328-521).  align_kernel.hip runs every read as a lane-private state machine out of a 75 KB
// slot in HBM and lets the whole wave solve DP problems one at a time; here each phase has the mapping that suits it:
//
//   chain_plan_kernel   one wavefront per read, the read's seeds / anchors / chaining arrays in LDS: frequency filter, anchors,
//                       the libstdc++ sort emulation, chaining DP, backtracking (chain.hpp:221-438), then the chain-selection
//                       loop run ahead of its DP results (which chains get scored does not depend on the scores except in one
//                       rare case that is detected later), and for every chain to score the problems of fill_chain
//                       (aligner_ksw2.hpp:2782-2979) appended to a task list binned by query length.  A ~1 KB plan per read is
//                       all that goes to HBM.
//   dp_lane_kernel      ksw_extz2_sse, one LANE per problem: the target (<= 104 rows) lives in registers, the kernel walks the
//                       query, 64 independent problems per wavefront with no cross-lane traffic and no LDS in the recurrence;
//                       direction bytes (right-aligned gap rule) stream to HBM, coalesced across the 64 lanes.  Scores do not
//                       depend on the gap-placement rule, so one pass serves both the score-only and the CIGAR call of chain_score.
//   select_kernel       one lane per read: consumes the results in the reference's order (fill_chain's score arithmetic,
//                       validity, lift, the best_scores / alternative-hit bookkeeping), picks the final chain.
//   traceback_kernel    one lane per problem of a final chain: ksw_backtrack over the stored direction bytes.
//   finish_kernel       one lane per read: CIGAR stitching (aligner_ksw2.hpp:3049-3108), then the shared record writer
//                       (lift, MD/NM, MAPQ, SAM text: ak_write_record).
//
// Anything outside the common case (more anchors / chains than the LDS arrays hold, overlapping anchors that need the global
// realignment, wildcard bases in a DP operand, problems larger than the register tile, the rare dependence of the selection loop
// on a score, an extension that would not reach the query end) is not approximated: the read is put on a list and
// align_kernel takes it as before.  Results are bit-identical either way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align_core.h"

#define AF_MAX_MEMS 64
#define AF_MAX_ANCH 256
#define AF_MAX_CHAINS 128
#define AF_MAX_CAND 16            // chains the selection loop scores for one read (the small LDS instance holds 8)
#define AF_PLAN_AN 96             // anchors of all those chains together (a chain's anchors are a slice of the plan's anchor pool: one long chain
                                  // or many short ones; the round-2 form gave every chain 6 slots, and 250 bp reads left the staged path for a 7th anchor)
#define AF_FIN_AN 16              // anchors / traced problems of the final chain that finish_wave_kernel stages in LDS (beyond: read from HBM)
#define AF_FIN_TB 12
#define AF_MAX_READ 512          // longest read of the staged path
#define AF_QCAP 256              // longest query of a DP task
#define AF_TB 104                // longest target of an extension / gap problem
#define AF_BLK 52                // target rows a lane keeps in registers at a time (H and F of a block: 104 registers, three waves per SIMD)
#define AF_LPASS (AF_TB / AF_BLK) // blocks of an extension / gap problem
#define AF_TS 32                 // target rows (one block) and longest query of the small tile (gap fills)
#define AF_GBLK 104               // register block of the global problems (overlapping anchors)
#define AF_GPASS 3                // their target blocks: up to AF_GPASS * AF_GBLK target rows
#define AF_NBIN 99               // 16 query-length bins of the large tile, 3 of the small tile, 16 of the banded extensions / gap fills, 16 of the global problems, 16 of the banded global
                                 // problems, 16 of the extensions / gap fills that may hold a wildcard base, 16 of such global problems
#define AF_BIN_SMALL 16u
#define AF_NSMALL 3u
#define AF_BIN_BAND 19u          // extensions and gap fills with a proven band: computed BEFORE global_task_kernel (a global problem's window comes from its chain's extensions)
#define AF_BIN_GLOBAL 35u
#define AF_BIN_GBAND 51u
#define AF_BIN_WILD 67u          // a wildcard base (N, any byte outside A / C / G / T) in an operand: scored by the kernels' WILDC instances, never banded
#define AF_BIN_GWILD 83u
#define AF_BANDW 16              // diagonals a banded global problem keeps in registers (dp_band_kernel)
#define AF_BIN_PROV ((uint32_t)AF_NBIN)      // one more queue: the large tile's problems before band_tasks_kernel has looked at them
#define AF_POS_BITS 25           // task_pos: a task's place in its bin's queue (a queue holds fewer than 2^25 entries), its bin (< 128) above
static_assert(AF_NBIN + 1 <= (1 << (32 - AF_POS_BITS)), "task_pos");
#define AF_TB_CIG 24             // CIGAR operations kept per traced problem
#define AF_FIN_CIG 96            // ... of a stitched alignment
#define AF_FIN_LCIG 160          // ... after lifting
#define AF_TXT_SHARDS 32
#define AF_NEG_INF (-0x40000000)

enum { AF_GAP_NONE = 0, AF_GAP_INS, AF_GAP_DEL0, AF_GAP_TASK, AF_GAP_1X1 };
enum { AF_ST_UNALIGNED = 0, AF_ST_CAND = 1, AF_ST_FALLBACK = 2, AF_ST_FINAL = 3, AF_ST_CHAINS = 4 /* chain_plan_kernel has left the read's chains for plan_kernel */ };

struct af_anchor_t {             // one anchor of a chain to score, and the gap between it and the next one (16 bytes)
    uint64_t occ;
    uint16_t len, idx;           // mem_t::len, mem_t::idx
    int16_t gap_val;             // AF_GAP_INS: the insertion's length; AF_GAP_DEL0 / AF_GAP_1X1: the score (closed forms, aligner_ksw2.hpp:2917-2945)
    uint8_t gap_kind, pad;
};
__device__ __forceinline__ int32_t af_ins_score(const ac_params_t& P, uint64_t l) {      // -min(gapo + l*gape, gapo2 + l*gape2), size_t arithmetic
    const uint64_t c1 = (uint64_t)(int64_t)P.gapo + l * (uint64_t)(int64_t)P.gape, c2 = (uint64_t)(int64_t)P.gapo2 + l * (uint64_t)(int64_t)P.gape2;
    return (int32_t)(0 - (c1 < c2 ? c1 : c2));
}
struct af_cand_t {               // one chain the selection loop scores (in loop order), 32 bytes
    int32_t chain_score;
    uint16_t chain_idx;
    uint8_t n_an, strand;
    uint32_t task0;              // its DP tasks: [left ext][right ext][gap fills in anchor order]
    uint8_t has_lc, has_rc, n_gap_tasks, overlap;      // overlap: anchors overlap, the chain is scored by one global problem (gtask) over the window the extensions give
    int32_t score;               // filled by select_kernel
    uint32_t gtask;
    uint16_t an0, pad;           // its anchors, left to right: an[an0 .. an0 + n_an) of the plan
    uint32_t pad2;
};
static_assert(sizeof(af_cand_t) == 32, "af_cand_t");
// the plan's 40-byte header: status, final chain, strand, traced problems, window - all finish_wave_kernel needs to start its fetches
#define AF_PLAN_HEADER \
    uint8_t status, n_cand; uint16_t n_chains; \
    int32_t min_score; \
    uint8_t final_cand, n_alt; uint16_t pad;      /* pad: strand of the final chain | its traced problems << 8 (select_kernel) */ \
    int32_t score2; \
    uint64_t ref_pos, ref_len;   /* window of the final chain */ \
    uint32_t tb0, pad2;          /* first traced problem of the final chain: [left ext][right ext][gap fills]; pad2: an0 | n_an << 16 of the final chain */
struct af_plan_t {
    AF_PLAN_HEADER
    af_cand_t cand[AF_MAX_CAND];
    af_anchor_t an[AF_PLAN_AN];
    uint64_t alt_pos[AF_MAX_CAND];      // (select_kernel)
    int32_t alt_score[AF_MAX_CAND];
};
static_assert(offsetof(af_plan_t, cand) == 40, "af_plan_t header");
// the plan while chain_plan_kernel builds it in LDS: NC chains to score, NA anchors, NT DP problems
template <int NC, int NA, int NT>
struct af_plan_lds_t {
    AF_PLAN_HEADER
    af_cand_t cand[NC];
    af_anchor_t an[NA];
    moni_dp_task_t tasks[NT];
    uint32_t n_an;               // anchors in use
};
struct af_ctab_t { int32_t score; uint16_t an0; uint8_t cnt, strand; uint64_t left_ref; };      // one chain of a read, in the order of chain.hpp:402's sort: its anchors are an[an0 .. an0 + cnt) of the read's plan
static_assert(sizeof(af_ctab_t) == 16, "af_ctab_t");
#define AF_CTAB 48               // chains per read in the table (the small LDS instance's capacity)
struct af_res_t { int32_t mqe, mqe_t, score, flags; };      // flags: 1 = a wildcard base in an operand (not computed)
struct af_chunk_t { uint32_t bin, start, n, qhi; uint64_t dir_off; };
struct af_tb_t { uint32_t n_ops; uint32_t ops[AF_TB_CIG]; };      // raw backtrack order (end -> start); n_ops = ~0u: did not fit

struct pe_sel_t;
struct af_args_t {
    ak_args_t A;                             // reads / text / seeds / record pools: as align_kernel
    uint32_t pe;                             // paired-end launch (pe_fast.hip): a plan belongs to a PAIR, plan r_in = pair A.read_lo + r_in = reads 2 * (A.read_lo + r_in) + mate;
                                             // a chain to score (af_cand_t) is one mate's share of a paired chain, its mate in bit 0 of af_cand_t::pad
    pe_sel_t* pe_sel;                        // per pair: what pe_select_kernel decided (pe_fast.hip)
    af_plan_t* plans;                        // per read of the launch
    moni_dp_task_t* tasks; uint32_t task_cap;    // slots: AF_MAX_TASKS_READ per read of the launch (chain_plan_kernel writes a read's tasks to its own slots: no shared
                                             // counter - one address takes only ~70 M atomics/s, one per read was the kernel's bound), then the global problems
    uint8_t* ntasks;                         // per read: how many of its slots are in use
    af_res_t* res;
    uint32_t* bin_q; uint32_t bin_cap;       // AF_NBIN queues of task indices
    uint32_t* task_pos;                      // where a task sits: position in its bin's queue | bin << AF_POS_BITS
    af_chunk_t* chunks; uint32_t chunk_cap;  // 64-task chunks of the two tiles: large first
    uint8_t* dirs; uint64_t dirs_cap;
    uint32_t* tb_task; af_tb_t* tb; uint32_t tb_cap;      // problems to trace
    uint32_t* fb_list; uint32_t* fb_n;       // reads handed to align_kernel and their number: per sub-batch, not per buffer set (align_kernel reads them on its own stream
                                             // while the set's next sub-batch is already running)
    const uint64_t* pat; const moni_u64x2* blk;      // the seeding stage's pattern workspace of the resident batch (seed_core.h: 2-bit code and mask words of every task, in read order)
    const uint64_t* text2; const uint32_t* exc; uint32_t exc_sh;      // the 2-bit text and its exception bitmap (seed_core.h: mem_fast_t)
    const uint8_t* pflag;                    // per seeding task (2 read + strand): the pattern holds a byte outside A / C / G / T
    af_ctab_t* ctab;                         // AF_CTAB per read of the launch: LEVEL 0's chains, for plan_kernel (nullptr: the LEVEL-0 instance plans by itself)
    uint32_t* list0; uint32_t* big_list; uint32_t* huge_list; // reads (indices in the launch) of the small / the large / the largest instance of chain_plan_kernel (classify_kernel)
    unsigned long long* txt_cur;             // AF_TXT_SHARDS cursors (one per 64 bytes) of the text pool's shard regions: a device-scope atomic on ONE address
    uint64_t txt_shard_words;                // sustains only ~50 M/s; region s + 1 of the pool (txt_shard_words each) belongs to shard s, region 0 to cursors[15]
    uint8_t* fin_scratch; uint64_t fin_stride;            // per finish lane: stitched CIGAR, lifted CIGAR, MD and text staging
    uint64_t* bnd;                           // per resident DP wave: (H, E) of a target block's last row for every query position (global problems)
    uint32_t* ctr;                           // AF_NCTR counters, see the AFC_* indices
    unsigned long long* prof;                // AF_PROFILE builds: wave cycles per phase
    uint32_t l0_mm, l0_ma;                   // capacities (seeds, anchors) of the instance the launch uses for LEVEL 0 (classify_kernel)
    uint32_t wave_max;                       // a group of the WILD / GLOBAL / GWILD queues with at most this many chunks is computed by dp_wave_kernel (af_chunk_kernel decides; MONI_AF_WAVE_MAX)
    uint32_t dbg;                            // AF_PROFILE builds: 1 = no direction stores, 2 = no DP rows (timing experiments; results are wrong)
};
#ifdef AF_PROFILE
#define AF_STAMP(var) const long long var = clock64()
#define AF_PROF(G, slot, t0, t1) do { if (threadIdx.x == 0 && (G).prof) atomicAdd(&(G).prof[slot], (unsigned long long)((t1) - (t0))); } while (0)
#else
#define AF_STAMP(var) do {} while (0)
#define AF_PROF(G, slot, t0, t1) do {} while (0)
#endif
enum { AFC_TASKS = 0, AFC_FALLBACK = 1, AFC_TRACED = 2, AFC_READ_CUR = 3, AFC_DIRS_OVF = 4, AFC_CELLS = 6 /* 64 bit */, AFC_DIROFF = 8 /* 64 bit */,
       AFC_NCHUNKS = 10 /* + group (7) */, AFC_CURSOR = 18 /* + group (7) */, AFC_BIG = 26, AFC_BIG_CUR = 27, AFC_HUGE = 28, AFC_HUGE_CUR = 29, AFC_L0 = 30 /* reads of the small instance's list */,
       AFC_NT = 31 /* DP problems queued by bin_tasks_kernel */, AFC_WHY = 32 /* + reason (12) */, AFC_RBYTES = 44 /* 64 bit: text bytes of the DP targets, the R of SURVEY.md 8(d) */,
       AFC_QBYTES = 46 /* 64 bit: read bytes of the DP queries */,
       AFC_SLOTS = 48 /* 64 bit: cell slots the DP kernels computed (query positions of the chunk's longest problem x target rows of its passes (or diagonals of its band) x 128 problems) */,
       AFC_CUTCELLS = 50 /* 64 bit: cells of the problems after an extension's target rows are cut (af_build_cand) and a problem is banded (af_tile_band2, af_global_band); AFC_CELLS counts them as the reference poses them */,
       AFC_BANDH = 52 /* + min(W / 4, 13): global problems by the width of the band of diagonals their optimal paths can touch (af_global_band) */,
       AFC_BINS = 68 /* + bin */, AFC_WMODE = 172 /* + group (7): dp_wave_kernel computes the group, not dp_lane_kernel */, AF_NCTR = 192 };
static_assert(AFC_BINS + AF_NBIN + 1 <= AFC_WMODE && AFC_WMODE + 7 <= AF_NCTR && AFC_WHY + 12 <= AFC_RBYTES && AFC_BANDH + 14 <= AFC_BINS && AFC_NCHUNKS + 7 <= AFC_CURSOR && AFC_CURSOR + 7 <= AFC_BIG, "counter layout");
enum { AF_GRP_LARGE = 0, AF_GRP_SMALL = 1, AF_GRP_BAND = 2, AF_GRP_WILD = 3, AF_GRP_GLOBAL = 4, AF_GRP_GBAND = 5, AF_GRP_GWILD = 6 };

// ------------------------------------------------------------------------------------------------------------------------------
// chain_plan_kernel
// ------------------------------------------------------------------------------------------------------------------------------
struct af_mem_t { uint64_t occ_off; uint32_t nocc; uint16_t len, idx, rpos; uint8_t mate, pad; };
struct af_chain_t { int32_t score; uint16_t off, cnt; uint32_t mate; };      // mate: of the chain's start anchor (bits 0-1) | the chain holds anchors of both mates << 8
struct af_start_t { int32_t f, j; };
struct af_left_t { uint64_t ref; int64_t score; };
#define AF_MAX_TASKS_READ 64       // DP task slots of one read in HBM (beyond: align_kernel)
// LDS of one read.  MA / MC / MM: capacities for anchors, chains, seeds.  Two instances are launched: a small one that most reads
// fit (more reads in flight per CU: the kernel is bound by the latency of its serial parts), and a large one for the reads that
// overflow it.  Arrays that are dead by the time the plan is written share their space with it.
template <int MA_, int MC_, int MM_, int NC_, int NA_, int NT_, int PE_ = 0, int SEC_ = 0>
struct af_wave_tt {
    static constexpr int MA = MA_, MC = MC_, MM = MM_, NC = NC_, NA = NA_, NT = NT_;
    static constexpr bool SEC = SEC_ != 0;      // -Z: the second track of find_chains_secondary (chain.hpp:442-727): f2 / msc2 / p2 / t2 per anchor, twice the pool
    static_assert(MC_ <= MA_, "the sorts of the chain starts and of the chains use per-anchor arrays as scratch");
    static constexpr int NSTACK = MA_ <= 128 ? 16 : MA_ <= 512 ? 20 : 24;      // pending partitions of the introsort emulation: at most 2 * floor(log2 n) + 1
    af_mem_t mem[MM_];
    uint64_t anch[MA_];                  // x (reference end, 40 bits) | mem << 40
    af_chain_t chains[MC_];
    uint16_t pool[(MA_ + MC_) * (SEC_ ? 2 : 1)];
    uint32_t n_chains_sh, status_sh, n_tasks;
    union {
        struct {                         // chaining (dead once af_chain has returned)
            lsort::frame stack[NSTACK];
            int32_t f[MA_], msc[MA_];
            int16_t p[MA_], t[MA_];
            af_start_t starts[MC_];
            uint16_t run_start[MA_ + 1];
            uint16_t s_off[MC_], s_cnt[MC_];       // per sorted start: where its chain's anchors are in the pool, how many (0: chain dropped)
            int32_t f2[SEC_ ? MA_ : 1], msc2[SEC_ ? MA_ : 1];
            int16_t p2[SEC_ ? MA_ : 1], t2[SEC_ ? MA_ : 1];
        };
        struct {                         // the selection loop and the plan
            af_plan_lds_t<NC_, NA_, NT_> plan;      // the plan and its tasks
            uint64_t left_ref[MC_];              // check_left_MEM's lifted coordinate of every chain (lanes in parallel: each lift is a chain of dependent loads)
            uint64_t left_ref2[PE_ ? MC_ : 1];   // paired-end: check_paired_left_MEM's coordinate of mate 2 (left_ref: of mate 1)
            uint16_t left_idx[MC_];              // the chains check_left_MEM has recorded
            int64_t diff[8];
        };
    };
};
static_assert(AF_MAX_TASKS_READ <= 255, "ntasks is a byte");
// (an instance needs MC <= MA: the sorts of the chain starts and of the chains take per-anchor arrays as their scratch)
typedef af_wave_tt<96, 48, 24, 8, 32, 32> af_wave_small_t;                                                   // most reads: 8 waves per SIMD
typedef af_wave_tt<192, 96, 32, 16, 64, 48> af_wave_mid_t;            // LEVEL 0 for reads of more than 200 bases (250 bp x 21 sequences: ~120 anchors per read, p99 210: 27 % of the reads fit the small instance, 98.8 % this one): 4 waves per SIMD (a 160-anchor one at 5 waves per SIMD: 7.70 against 7.88 M reads/s)
typedef af_wave_tt<AF_MAX_ANCH, AF_MAX_CHAINS, AF_MAX_MEMS, AF_MAX_CAND, AF_PLAN_AN, AF_MAX_TASKS_READ> af_wave_t;   // reads that overflow it
typedef af_wave_tt<2048, 1024, 256, AF_MAX_CAND, AF_PLAN_AN, AF_MAX_TASKS_READ> af_wave_huge_t;             // repeat-rich reads (hundreds of occurrences per seed): one wave per CU

// why a read left the staged path (counted in ctr[AFC_WHY + reason])
enum { AF_WHY_LONG = 0, AF_WHY_ANCHORS, AF_WHY_CHAINS, AF_WHY_CANDS, AF_WHY_CHAIN_LEN, AF_WHY_TASK_SIZE, AF_WHY_OVERLAP, AF_WHY_WILDCARD, AF_WHY_LOOP, AF_WHY_REACH_END,
       AF_WHY_CAPACITY, AF_WHY_CIGAR, AF_WHY_N };
#define AF_FALLBACK(G, why) (atomicAdd(&(G).ctr[AFC_WHY + (why)], 1u), (uint32_t)AF_ST_FALLBACK)
#define AF_X(a) ((a) & MONI_POS_MASK_)
#define MONI_POS_MASK_ ((1ull << 40) - 1)

// Bins of the DP problems.  A chunk of 128 problems runs as many query positions as its longest problem has, so bins are narrow where problems are
// many: the large tile's upper bounds are 8, 13 (an extension of up to 13 bases needs 50 target rows: one pass of the 52-row block, af_build_cand),
// 16, 24, 32, then every 16 up to 160, then every 32 up to AF_QCAP; the small tile's 8, 16, 32.
__device__ __forceinline__ uint32_t af_large_bin(int q) {
    if (q <= 16) return q <= 8 ? 0u : q <= 13 ? 1u : 2u;
    if (q <= 32) return 3u + (uint32_t)((q - 17) >> 3);
    if (q <= 160) return 5u + (uint32_t)((q - 33) >> 4);
    return 13u + (uint32_t)((q - 161) >> 5);
}
__device__ __forceinline__ uint32_t af_large_qhi(uint32_t b) { return b < 3 ? (b == 0 ? 8u : b == 1 ? 13u : 16u) : b < 5 ? 24u + 8u * (b - 3) : b < 13 ? 48u + 16u * (b - 5) : 192u + 32u * (b - 13); }
static_assert(AF_QCAP == 256, "af_large_bin covers query lengths up to 256");
__device__ __forceinline__ uint32_t af_bin_of(int qlen, int tlen) {
    return (qlen <= AF_TS && tlen <= AF_TS) ? AF_BIN_SMALL + (qlen <= 8 ? 0u : qlen <= 16 ? 1u : 2u) : af_large_bin(qlen);
}
__device__ __forceinline__ uint32_t af_grp_of_bin(uint32_t bin) {
    return bin < AF_BIN_SMALL ? AF_GRP_LARGE : bin < AF_BIN_BAND ? AF_GRP_SMALL : bin < AF_BIN_GLOBAL ? AF_GRP_BAND : bin < AF_BIN_GBAND ? AF_GRP_GLOBAL : bin < AF_BIN_WILD ? AF_GRP_GBAND
         : bin < AF_BIN_GWILD ? AF_GRP_WILD : AF_GRP_GWILD;
}
__device__ __forceinline__ uint32_t af_grp_b0(uint32_t grp) {
    return grp == AF_GRP_LARGE ? 0u : grp == AF_GRP_SMALL ? AF_BIN_SMALL : grp == AF_GRP_BAND ? AF_BIN_BAND : grp == AF_GRP_GLOBAL ? AF_BIN_GLOBAL : grp == AF_GRP_GBAND ? AF_BIN_GBAND
         : grp == AF_GRP_WILD ? AF_BIN_WILD : AF_BIN_GWILD;
}
__device__ __forceinline__ uint32_t af_grp_b1(uint32_t grp) {
    return grp == AF_GRP_LARGE ? AF_BIN_SMALL : grp == AF_GRP_SMALL ? AF_BIN_BAND : grp == AF_GRP_BAND ? AF_BIN_GLOBAL : grp == AF_GRP_GLOBAL ? AF_BIN_GBAND : grp == AF_GRP_GBAND ? AF_BIN_WILD
         : grp == AF_GRP_WILD ? AF_BIN_GWILD : (uint32_t)AF_NBIN;
}
__device__ __forceinline__ uint32_t af_bin_qhi(uint32_t bin) {
    return bin < AF_BIN_SMALL ? af_large_qhi(bin) : bin < AF_BIN_BAND ? (8u << (bin - AF_BIN_SMALL)) : bin < AF_BIN_GLOBAL ? af_large_qhi(bin - AF_BIN_BAND)
         : bin < AF_BIN_GBAND ? (bin - AF_BIN_GLOBAL + 1u) * 16u : bin < AF_BIN_WILD ? af_large_qhi(bin - AF_BIN_GBAND) : bin < AF_BIN_GWILD ? af_large_qhi(bin - AF_BIN_WILD)
         : (bin - AF_BIN_GWILD + 1u) * 16u;
}
// target rows (or diagonals) per block and blocks of a group's problems: what a chunk's direction bits are laid out by
__device__ __forceinline__ uint32_t af_grp_tb(uint32_t grp) {
    return grp == AF_GRP_SMALL ? AF_TS : (grp == AF_GRP_GLOBAL || grp == AF_GRP_GWILD) ? AF_GBLK : (grp == AF_GRP_BAND || grp == AF_GRP_GBAND) ? AF_BANDW : AF_BLK;
}
__device__ __forceinline__ uint32_t af_grp_np(uint32_t grp) { return (grp == AF_GRP_GLOBAL || grp == AF_GRP_GWILD) ? AF_GPASS : (grp == AF_GRP_LARGE || grp == AF_GRP_WILD) ? AF_LPASS : 1; }
__device__ __forceinline__ bool af_bin_banded(uint32_t bin) { return (bin >= AF_BIN_BAND && bin < AF_BIN_GLOBAL) || (bin >= AF_BIN_GBAND && bin < AF_BIN_WILD); }
// the text blocks a target lies in hold a byte outside A / C / G / T (the exception bitmap of the 2-bit text): the problem may hold a wildcard
__device__ __forceinline__ bool af_target_exc(const af_args_t& G, const moni_dp_task_t& T) {
    if (!G.exc || T.tlen <= 0) return true;
    const uint64_t lo = (T.reserved & DP_T_REV) ? (T.t_off + 1 >= (uint64_t)T.tlen ? T.t_off + 1 - (uint64_t)T.tlen : 0ull) : T.t_off;
    const uint64_t hi = (T.reserved & DP_T_REV) ? T.t_off : T.t_off + (uint64_t)T.tlen - 1;
    bool x = false;
    for (uint64_t b = lo >> G.exc_sh; b <= (hi >> G.exc_sh); ++b) x = x || ((G.exc[b >> 5] >> (b & 31u)) & 1u);
    return x;
}

// The lanes that work on one read: the whole wavefront (GW = 64) or a GROUP of GW consecutive lanes (GW = 16: four reads per wavefront, each with
// its own LDS state).  chain_plan_kernel issues mostly one-lane instructions - the selection loop, the backtracking, a lane per run of anchors - and
// with one read per wavefront 63 of 64 lanes idle while the SIMD's issue slots are the bound (8 waves per SIMD x 0.11 VALU-active = 0.89 of the issue
// cycles); with four reads per wavefront those sections run four wide.  Control flow is uniform within a group and may diverge between the groups of
// a wavefront: a ballot is cut down to the group's own lanes (lanes of another group that are not at the same instruction read as 0 and are not
// looked at), shuffles stay inside the group, and __syncthreads() - one wavefront per workgroup - only orders the wave's own LDS traffic.
template <int GW>
struct af_grp_t {
    static constexpr int W = GW;
    static_assert(GW == 64 || GW == 32 || GW == 16, "group width");
    int lane, base;          // lane within the group; the group's first lane in the wavefront
    __device__ __forceinline__ af_grp_t() : lane((int)(threadIdx.x & (GW - 1))), base((int)(threadIdx.x & 63u & ~(unsigned)(GW - 1))) {}
    __device__ __forceinline__ unsigned long long ballot(bool p) const {
        const unsigned long long b = __ballot(p);
        if (GW == 64) return b;
        return (b >> base) & ((1ull << (GW & 63)) - 1ull);
    }
    __device__ __forceinline__ unsigned long long lt_mask() const { return (1ull << lane) - 1ull; }
    template <class T> __device__ __forceinline__ T shfl(T v, int src) const { return __shfl(v, base + src); }
    __device__ __forceinline__ int id() const { return base / GW; }
};

// std::sort of n <= 16 elements is one insertion sort, i.e. a stable sort: every lane places one element by its stable rank
template <class T, class Less>
__device__ __forceinline__ void af_small_sort(T* a, uint32_t n, Less less, int lane) {
    T v; uint32_t rank = 0;
    if ((uint32_t)lane < n) { v = a[lane]; for (uint32_t k = 0; k < n; ++k) { const T w = a[k]; rank += (less(w, v) || (!less(v, w) && k < (uint32_t)lane)) ? 1u : 0u; } }
    __syncthreads();
    if ((uint32_t)lane < n) a[rank] = v;
    __syncthreads();
}

// std::sort of the anchors by reference end, by the whole wave, element for element what libstdc++'s introsort leaves (sort_emul.h
// is the serial form; ties between anchors are common - a MEM's right half ends where the MEM ends - and their order reaches the SAM):
//   * the introsort loop keeps its order of partitions (explicit stack), but a partition is done by all lanes at once: the unguarded
//     Hoare partition swaps the k-th element from the left that is not below the pivot with the k-th from the right that is not above
//     it for as long as the former lies left of the latter - ranks that ballots give without walking;
//   * std::__final_insertion_sort (guarded for the first 16, unguarded after) is a stable sort: every lane places its elements by
//     stable rank;
//   * the depth-limit fallback (heap sort of a range) stays serial: it needs 2 * log2(n) unlucky partitions in a row.
// NC: elements per lane (n <= GW * NC, GW the lanes of the read's group).  ipos / jpos: scratch of n entries each.
// key(x): the integer the elements are ordered by (ascending).
// piece: scratch of n 32-bit entries - the leaf range (first | last << 16) of the introsort loop that holds an element.  Leaves lie in order (what a Hoare
// partition leaves on its left is not above what it leaves on its right), a stable sort never moves an element in front of an equal one, so the final
// insertion sort keeps every element inside its leaf: its stable rank there (<= 16 comparisons) is its place - not its rank in the whole array (n of them).
template <int NC, class GT, class T, class Key>
__device__ __forceinline__ void af_wave_sort(const GT& g, T* a, uint32_t n, lsort::frame* st, uint16_t* ipos, uint16_t* jpos, uint32_t* piece, Key key) {
    constexpr uint32_t GW = GT::W;
    const int lane = g.lane;
    auto less = [&](const T& x, const T& y) { return key(x) < key(y); };
    const unsigned long long lt_mask = g.lt_mask();
    if (n > 16) {
        int lg = 0;
        for (uint32_t v = n; v > 1; v >>= 1) ++lg;
        int sp = 0;
        if (lane == 0) st[0] = lsort::frame{0, (int)n, lg * 2};
        sp = 1;
        __syncthreads();
        while (sp > 0) {
            const lsort::frame f = st[--sp];
            __syncthreads();
            long first = f.first, last = f.last, depth = f.depth;
            while (last - first > 16) {
                if (depth == 0) { if (lane == 0) lsort::heap_sort(a, first, last, less); __syncthreads(); break; }
                --depth;
                const long mid = first + (last - first) / 2;
                if (lane == 0) lsort::move_median_to_first(a, first, first + 1, mid, last - 1, less);
                __syncthreads();
                // ---- unguarded_partition(a, first + 1, last, first) ----
                const auto pk = key(a[first]);
                const uint32_t lo = (uint32_t)first + 1, hi = (uint32_t)last;
                T v[NC]; bool fa[NC], fb[NC]; uint32_t ra[NC], rb[NC];
                uint32_t nA = 0, nB = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {          // ranks from the left among the elements not below the pivot
                    const uint32_t i = lo + (uint32_t)lane + GW * c;
                    fa[c] = fb[c] = false;
                    if (i < hi) { v[c] = a[i]; const auto k = key(v[c]); fa[c] = !(k < pk); fb[c] = !(pk < k); }
                    const unsigned long long ba = g.ballot(fa[c]);
                    ra[c] = nA + (uint32_t)__popcll(ba & lt_mask);
                    nA += (uint32_t)__popcll(ba);
                }
#pragma unroll
                for (int c = NC - 1; c >= 0; --c) {     // ranks from the right among the elements not above it
                    const unsigned long long bb = g.ballot(fb[c]);
                    const unsigned long long gt_mask = ~((2ull << lane) - 1ull);
                    rb[c] = nB + (uint32_t)__popcll(bb & gt_mask);
                    nB += (uint32_t)__popcll(bb);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t i = lo + (uint32_t)lane + GW * c;
                    if (fa[c]) ipos[ra[c]] = (uint16_t)i;
                    if (fb[c]) jpos[rb[c]] = (uint16_t)i;
                }
                __syncthreads();
                uint32_t K = 0;                           // swaps: the pairs whose left element lies left of the right one (a prefix of the ranks)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t i = lo + (uint32_t)lane + GW * c;
                    const bool sw = fa[c] && ra[c] < nB && i < (uint32_t)jpos[ra[c]];
                    K += (uint32_t)__popcll(g.ballot(sw));
                }
                const uint32_t iK = K < nA ? (uint32_t)ipos[K] : 0xFFFFFFFFu, jK1 = K > 0 ? (uint32_t)jpos[K - 1] : 0xFFFFFFFFu;
                const long cut = (long)(iK < jK1 ? iK : jK1);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (fa[c] && ra[c] < K) a[jpos[ra[c]]] = v[c];
                    if (fb[c] && rb[c] < K) a[ipos[rb[c]]] = v[c];
                }
                __syncthreads();
                // the reference recurses into [cut, last) first and then loops on [first, cut): disjoint ranges, either order gives the same array
                if (lane == 0) st[sp] = lsort::frame{(int)cut, (int)last, (int)depth};
                ++sp;
                last = cut;
                __syncthreads();
            }
            // [first, last) is a leaf: at most 16 elements, or a range the heap sort has put in order
            for (uint32_t i = (uint32_t)first + (uint32_t)lane; i < (uint32_t)last; i += GW) piece[i] = (uint32_t)first | ((uint32_t)last << 16);
        }
    } else {
        for (uint32_t i = (uint32_t)lane; i < n; i += GW) piece[i] = n << 16;
    }
    __syncthreads();
    // ---- std::__final_insertion_sort == a stable sort of what the loop left: every element by its stable rank inside its leaf ----
    T v[NC]; uint32_t rk[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t i = (uint32_t)lane + GW * c;
        rk[c] = 0;
        if (i < n) {
            v[c] = a[i];
            const auto kx = key(v[c]);
            const uint32_t pc = piece[i], p0 = pc & 0xFFFFu, p1 = pc >> 16;
            rk[c] = p0;
            for (uint32_t k = p0; k < p1; ++k) { const auto kk = key(a[k]); rk[c] += (kk < kx || (kk == kx && k < i)) ? 1u : 0u; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) if ((uint32_t)lane + GW * c < n) a[rk[c]] = v[c];
    __syncthreads();
}

// The whole wave: anchors -> chains (chain.hpp:221-438) over the wave's LDS arrays.  The sorts are the libstdc++ emulation where
// ties exist (one lane; short arrays by stable rank); the chaining DP and the backtracking run one lane per RUN of anchors: a run
// is a maximal stretch of the sorted anchors whose consecutive reference ends are at most max_dist_x apart, pairs from different
// runs never pass the distance test of chain.hpp:300 (or are skipped for their mates before it), so runs share nothing but the
// lower bound `lb`, which only ever excludes anchors that are too far anyway.  Returns the plan status (uniform).
template <class WT, class GT = af_grp_t<64>>
__device__ __forceinline__ uint32_t af_chain(const af_args_t& G, WT& L, uint32_t na, float avg_mem_length, const GT g = GT(), const bool sec_on = false) {
    const ac_params_t& P = G.A.P;
    constexpr uint32_t GW = GT::W;
    const int lane = g.lane;
    // ---- std::sort of the anchors by reference end (chain.hpp:246) ----
    AF_STAMP(s0);
#if defined(AF_PROFILE) || defined(AF_CUTS)
    if (G.dbg & 32) { /* timing experiment: the anchors stay unsorted */ } else
#endif
    if (G.dbg & 64) {          // MONI_AF_DBG=64 (any build): the serial form, for cross-checking the wave's
        if (na <= 16) af_small_sort(L.anch, na, [](const uint64_t& a, const uint64_t& b) { return AF_X(a) < AF_X(b); }, lane);
        else { if (lane == 0) lsort::sort(L.anch, (long)na, [](const uint64_t& a, const uint64_t& b) { return AF_X(a) < AF_X(b); }, L.stack); __syncthreads(); }
    } else {
        af_wave_sort<(WT::MA + GT::W - 1) / GT::W>(g, L.anch, na, L.stack, L.run_start, reinterpret_cast<uint16_t*>(L.p), reinterpret_cast<uint32_t*>(L.f), [](const uint64_t& x) { return AF_X(x); });
        for (uint32_t i = lane; i < na; i += GW) L.p[i] = 0;        // (scratch of the sort)
        __syncthreads();
    }
    AF_STAMP(s1); AF_PROF(G, 5, s0, s1);
    // ---- runs ----
    uint32_t n_runs = 0;
    for (uint32_t i0 = 0; i0 < na; i0 += GW) {
        const uint32_t i = i0 + lane;
        const bool brk = i < na && (i == 0 || (long long)AF_X(L.anch[i]) > (long long)AF_X(L.anch[i - 1]) + P.max_dist_x);
        const unsigned long long bal = g.ballot(brk);
        if (brk) L.run_start[n_runs + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)i;
        n_runs += (uint32_t)__popcll(bal);
    }
    if (lane == 0) L.run_start[n_runs] = (uint16_t)na;
    __syncthreads();
#if defined(AF_PROFILE) || defined(AF_CUTS)
    if (G.dbg & 128) return AF_ST_UNALIGNED;          // timing experiment: stop after the sort
#endif
    // ---- chaining DP (chain.hpp:278-362), one lane per run ----
    for (uint32_t k = lane; k < n_runs; k += GW) {
        const uint32_t fb = L.run_start[k], fe = L.run_start[k + 1];
        long long lb = fb;
        for (uint32_t i = fb; i < fe; ++i) {
            const uint64_t aw = L.anch[i];
            const af_mem_t mi = L.mem[aw >> 40];
            const long long x_i = (long long)AF_X(aw), y_i = mi.rpos, w_i = mi.len;
            const uint32_t mate_i = mi.mate;
            long long max_f = w_i, max_j = -1, max_sec_f = w_i, max_sec_j = -1;
            size_t n_pred = 0;
            if ((size_t)i - (size_t)lb > (size_t)P.max_iter) lb = (long long)i - P.max_iter;
            for (long long j = (long long)i - 1; j >= lb; --j) {
                const uint64_t awj = L.anch[j];
                const af_mem_t mj = L.mem[awj >> 40];
                const long long x_j = (long long)AF_X(awj), y_j = mj.rpos;
                const uint32_t mate_j = mj.mate;
                if (mate_i != mate_j && ((mate_i ^ mate_j) != 3)) continue;
                if (x_i > x_j + P.max_dist_x) { lb = j; continue; }
                const long long x_d = x_i - x_j, y_d = y_i - y_j;
                const int32_t l = (int32_t)(y_d > x_d ? (y_d - x_d) : (x_d - y_d));
                const uint32_t ilog_l = l > 0 ? (uint32_t)(31 - __clz(l)) : 0;
                if (mate_i == mate_j && (y_j >= y_i || y_d > P.max_dist_y)) continue;
                const long long mn = y_d < x_d ? y_d : x_d;
                const long long alpha = mn < w_i ? mn : w_i;
                long long beta = 0;
                if (mate_i != mate_j) {
                    if (x_d == 0) ++beta;
                    else { const int c_lin = (int)(l * .01 * avg_mem_length); beta = c_lin < (long long)ilog_l ? c_lin : (long long)ilog_l; }
                } else {
                    beta = l > 0 ? ((long long)(.01 * l * avg_mem_length) + ilog_l) >> 1 : 0;
                }
                const long long score = L.f[j] + (alpha - beta);
                if (score > max_f) { max_f = score; max_j = j; if (n_pred > 0) --n_pred; }
                else if (WT::SEC && sec_on && (long long)L.f2[j] + (alpha - beta) > max_sec_f) {          // chain.hpp:586-612: the best predecessor that is not on the primary chain of the best one
                    if (max_j >= 0) {
                        const uint64_t pos_j = (uint64_t)x_j - mj.len + 1;                                 // the occurrence the anchor stands for
                        bool uniq = true;
                        for (long long tmp = max_j; tmp >= 0; tmp = L.p[tmp]) { const uint64_t at = L.anch[tmp]; if (AF_X(at) - L.mem[at >> 40].len + 1 == pos_j) { uniq = false; break; } }
                        if (uniq) { max_sec_f = (long long)L.f2[j] + (alpha - beta); max_sec_j = j; }
                    }
                }
                else if ((size_t)(long long)L.t[j] == (size_t)i && (++n_pred > (size_t)P.max_pred)) break;
                if (L.p[j] > 0) L.t[L.p[j]] = (int16_t)i;
                if (WT::SEC && sec_on && L.p2[j] > 0) L.t2[L.p2[j]] = (int16_t)i;
            }
            L.f[i] = (int32_t)max_f; L.p[i] = (int16_t)max_j;
            L.msc[i] = (max_j >= 0 && L.msc[max_j] > max_f) ? L.msc[max_j] : (int32_t)max_f;
            if (WT::SEC && sec_on) { L.f2[i] = (int32_t)max_sec_f; L.p2[i] = (int16_t)max_sec_j; L.msc2[i] = (max_sec_j >= 0 && L.msc2[max_sec_j] > max_sec_f) ? L.msc2[max_sec_j] : (int32_t)max_sec_f; }
        }
    }
    __syncthreads();
    AF_STAMP(s2); AF_PROF(G, 6, s1, s2);
#if defined(AF_PROFILE) || defined(AF_CUTS)
    if (G.dbg & 256) return AF_ST_UNALIGNED;          // ... after the chaining DP
#endif
    // ---- chain ends, starts, backtracking, chains of ONE track (chain.hpp:115-200, 363-400; the second track: 640-700): its f / msc / p / t arrays, where its
    // chains' anchors go in the pool, and how its starts are ordered - std::greater<pair> (ties are identical pairs: any sort gives the reference's array, by
    // rank) or, with -Z, by score alone (chain_start_cmp, chain.hpp:663-668: ties in libstdc++'s order - the introsort emulation).  Returns false when the
    // instance's capacities do not hold the track.
    uint32_t n_chains = 0;
    auto track = [&](int32_t* F, int32_t* MSC, int16_t* PP, int16_t* TT, uint32_t pool0, bool by_score_alone, uint32_t& n_starts) -> bool {
        for (uint32_t i = lane; i < na; i += GW) TT[i] = 0;
        __syncthreads();
        for (uint32_t i = lane; i < na; i += GW) if (PP[i] >= 0) TT[PP[i]] = 1;
        __syncthreads();
        uint32_t ns = 0;
        for (uint32_t i0 = 0; i0 < na; i0 += GW) {
            const uint32_t i = i0 + lane;
            const bool is_end = i < na && TT[i] == 0 && MSC[i] > P.min_chain_score;
            af_start_t st; st.f = 0; st.j = 0;
            if (is_end) { uint32_t j = i; while (F[j] < MSC[j]) j = (uint32_t)PP[j]; st.f = F[j]; st.j = (int32_t)j; }
            const unsigned long long bal = g.ballot(is_end);
            const uint32_t at = ns + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (is_end && at < (uint32_t)WT::MC) L.starts[at] = st;
            ns += (uint32_t)__popcll(bal);
        }
        n_starts = ns;
        if (ns > (uint32_t)WT::MC) return false;          // does not fit this instance
        __syncthreads();
        if (ns == 0) return true;
        if (by_score_alone) {          // (t, s_off and this track's part of the pool are free: scratch of the sort)
            af_wave_sort<(WT::MC + GT::W - 1) / GT::W>(g, L.starts, ns, L.stack, reinterpret_cast<uint16_t*>(TT), L.s_off, reinterpret_cast<uint32_t*>(L.pool + pool0), [](const af_start_t& x) { return -(int64_t)x.f; });
        } else {
            af_start_t v[(WT::MC + GT::W - 1) / GT::W]; uint32_t rk[(WT::MC + GT::W - 1) / GT::W];
#pragma unroll
            for (int q = 0; q < (WT::MC + GT::W - 1) / GT::W; ++q) {
                const uint32_t s = (uint32_t)lane + GW * q;
                rk[q] = 0;
                if (s < ns) { v[q] = L.starts[s]; for (uint32_t k = 0; k < ns; ++k) { const af_start_t w = L.starts[k]; rk[q] += (w.f > v[q].f || (w.f == v[q].f && (w.j > v[q].j || (w.j == v[q].j && k < s)))) ? 1u : 0u; } }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (WT::MC + GT::W - 1) / GT::W; ++q) if ((uint32_t)lane + GW * q < ns) L.starts[rk[q]] = v[q];
            __syncthreads();
        }
        // backtracking (chain.hpp:166-200), one lane per run, every lane over the sorted starts of its run in order
        for (uint32_t i = lane; i < na; i += GW) TT[i] = 0;
        __syncthreads();
        for (uint32_t k = lane; k < n_runs; k += GW) {
            const uint32_t fb = L.run_start[k], fe = L.run_start[k + 1];
            uint32_t used = pool0 + fb;                           // a run's chains use at most one pool entry per anchor and one more per start
            for (uint32_t s = 0; s < ns; ++s) used += (uint32_t)L.starts[s].j < fb ? 1u : 0u;
            // every lane goes through the sorted starts looking for those of its run; the lanes look first and then walk their chains TOGETHER (a lane that walked
            // as soon as it had found a start did so alone - the others were at other starts: 13 runs, 13 walks one after the other, 1330 of the kernel's 4980
            // instructions per read; profiles/r04d)
            uint32_t s = 0;
            while (true) {
                while (s < ns && ((uint32_t)L.starts[s].j < fb || (uint32_t)L.starts[s].j >= fe)) ++s;
                if (s >= ns) break;
                const af_start_t st = L.starts[s];
                long long j = st.j;
                uint32_t cnt = 0;
                const uint32_t off = used;
                const uint32_t mate0 = L.mem[L.anch[j] >> 40].mate;
                uint32_t paired = 0;                              // anchors of both mates (chain.hpp:186): only the paired-end path looks at it
                do { paired |= L.mem[L.anch[j] >> 40].mate != mate0 ? 1u : 0u; L.pool[used++] = (uint16_t)j; cnt++; TT[j] = 1; j = PP[j]; } while (j >= 0 && TT[j] == 0);
                bool keep = false;
                if (j < 0) keep = (long long)cnt >= P.min_chain_length;
                else if ((long long)st.f - F[j] >= P.min_chain_score) keep = (long long)cnt >= P.min_chain_length;
                L.s_off[s] = (uint16_t)off; L.s_cnt[s] = (uint16_t)(keep ? (cnt | (paired << 15)) : 0u);          // (cnt <= MA < 2^15)
                ++s;
            }
        }
        __syncthreads();
        uint32_t kept = 0;
        for (uint32_t s0 = 0; s0 < ns; s0 += GW) {
            const uint32_t sx = s0 + lane;
            const bool keep = sx < ns && L.s_cnt[sx] > 0;
            const unsigned long long bal = g.ballot(keep);
            const uint32_t at = n_chains + kept + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (keep && at < (uint32_t)WT::MC) {
                af_chain_t c;
                c.score = L.starts[sx].f; c.mate = L.mem[L.anch[L.starts[sx].j] >> 40].mate | ((uint32_t)(L.s_cnt[sx] >> 15) << 8); c.off = L.s_off[sx]; c.cnt = L.s_cnt[sx] & 0x7FFFu;
                L.chains[at] = c;
            }
            kept += (uint32_t)__popcll(bal);
        }
        n_chains += kept;
        __syncthreads();
        return n_chains <= (uint32_t)WT::MC;
    };
    uint32_t ns1 = 0, ns2 = 0;
    if (!track(L.f, L.msc, L.p, L.t, 0u, WT::SEC && sec_on, ns1)) return 0xFFu;
    if (ns1 == 0) return AF_ST_UNALIGNED;          // (no chain start: not chained, whatever the second track holds - chain.hpp:640)
#if defined(AF_PROFILE) || defined(AF_CUTS)
    if (G.dbg & (512 | 1024)) return AF_ST_UNALIGNED;          // timing experiment: stop after the chains of the first track
#endif
    if (WT::SEC && sec_on) { if (!track(L.f2, L.msc2, L.p2, L.t2, (uint32_t)(WT::MA + WT::MC), true, ns2)) return 0xFFu; }
    __syncthreads();
    // std::sort of the chains by score (chain.hpp:402): ties
    if (G.dbg & 64) {
        if (n_chains <= 16) af_small_sort(L.chains, n_chains, [](const af_chain_t& x, const af_chain_t& y) { return x.score > y.score; }, lane);
        else { if (lane == 0) lsort::sort(L.chains, (long)n_chains, [](const af_chain_t& x, const af_chain_t& y) { return x.score > y.score; }, L.stack); __syncthreads(); }
    } else          // by the whole wave (run_start and p are free again: scratch)
        af_wave_sort<(WT::MC + GT::W - 1) / GT::W>(g, L.chains, n_chains, L.stack, L.run_start, reinterpret_cast<uint16_t*>(L.p), reinterpret_cast<uint32_t*>(L.f), [](const af_chain_t& x) { return -(int64_t)x.score; });
    if (lane == 0) L.n_chains_sh = n_chains;
    __syncthreads();
    AF_STAMP(s3); AF_PROF(G, 7, s2, s3);
    return AF_ST_CAND;
}

// query segment R[a .. a+len) of the strand-oriented read, optionally reversed (ac_qseg)
__device__ __forceinline__ void af_qseg(uint64_t off, uint32_t m, uint32_t strand, uint64_t a, uint64_t len, bool reversed, uint64_t& q_off, int& qmode) {
    if (!strand) { q_off = reversed ? off + a + len - 1 : off + a; qmode = reversed ? DP_Q_REV : 0; }
    else { q_off = reversed ? off + (m - (a + len)) : off + (m - 1 - a); qmode = DP_Q_COMP | (reversed ? 0 : DP_Q_REV); }
}
// The anchors of chain C.chain_idx whose mate bit (mem_t::mate & 1) is in `mates` (bit 0: mate 1, bit 1: mate 2; single-end: 3 = all), left to right, into
// the plan's pool from C.an0 on: at most `room` of them (the caller's share of the pool); returns how many there are (C.n_an = those written)
template <class WT>
__device__ __forceinline__ uint32_t af_cand_anchors(WT& L, af_cand_t& C, uint32_t mates, uint32_t room_or_0) {
    const af_chain_t ch = L.chains[C.chain_idx];
    af_anchor_t* const AN = L.plan.an + C.an0;
    uint32_t n = 0;
    for (uint32_t k = 0; k < ch.cnt; ++k) {                      // stored right to left (chain.hpp:166-200); fill_chain wants left to right
        const uint64_t aw = L.anch[L.pool[ch.off + ch.cnt - 1 - k]];
        const af_mem_t mk = L.mem[aw >> 40];
        if (!((mates >> (mk.mate & 1u)) & 1u)) continue;
        if (room_or_0 == 0 || n < room_or_0) {
            af_anchor_t A;
            A.occ = AF_X(aw) - mk.len + 1; A.len = mk.len; A.idx = mk.idx; A.gap_val = 0; A.gap_kind = AF_GAP_NONE; A.pad = 0;
            AN[n] = A;
            if (n == 0) C.strand = (mk.mate & 2) ? 1 : 0;
        }
        ++n;
    }
    return n;
}

// One chain to score (fill_chain, part 1: aligner_ksw2.hpp:2782-2979) by ONE lane: its anchors left to right into the plan's pool, the gaps between
// them classified (closed forms need no DP: a pure insertion, the "deletion" the reference scores with l = 0, one base against one base), and its DP
// problems [left extension][right extension][gap fills in anchor order].  MODE 0: count the problems and classify the gaps; MODE 1: write the problems from
// task0 on; MODE 2: both at once.  Returns the number of problems, or 0xFFFFFFFF when one is beyond the DP tiles (the read leaves the staged path).
template <int MODE, class WT>
__device__ __forceinline__ uint32_t af_build_cand(const af_args_t& G, WT& L, af_cand_t& C, uint64_t off, uint32_t m, uint32_t task0) {
    constexpr bool WRITE = MODE != 0, CLASSIFY = MODE != 1;          // MODE 0: count and classify; 1: write what MODE 0 classified; 2: both in one pass (plan_kernel)
    const ac_params_t& P = G.A.P;
    auto& PL = L.plan;
    struct { uint32_t cnt; } ch; ch.cnt = C.n_an;          // the anchors are in the plan's pool already (af_cand_anchors / the paired path's share of a chain)
    af_anchor_t* const AN = PL.an + C.an0;
    const uint64_t n_text = P.n_text, ext_len = P.ext_len;
    uint32_t nt = 0;
    bool bad = false;
    auto add = [&](uint64_t q_off, uint64_t qlen, int qmode, uint64_t t_off, uint64_t tlen, int tmode, int flag) {
        if (qlen == 0 || tlen == 0 || qlen > AF_QCAP || tlen > AF_TB) { bad = true; return; }
        if ((flag & DP_EZ_EXTZ_ONLY) && G.A.D.e > 0) {
            // An extension is asked for mqe = max_i H(i, qlen - 1), the first row that holds it and the traceback from there: rows that
            // cannot hold it need not be computed.  A path to (i, qlen - 1) with d = i + 1 - qlen > 0 takes at most qlen diagonal steps and deletes at
            // least d target bases: H <= qlen sc_mch - (qo + d e); the diagonal itself gives mqe >= H(qlen - 1, qlen - 1) >= qlen min(sc_mis, sc_N).  Rows
            // with d e > qlen (sc_mch - min(sc_mis, sc_N)) - qo lie strictly below that, and no cell of the kept rows depends on them: the target is cut
            // there (an extension of 10 bases against the reference's 100 target rows keeps 38).  The uncut length rides in the flag word for the counters.
            const int32_t worst = G.A.D.sc_mis < G.A.D.sc_N ? G.A.D.sc_mis : G.A.D.sc_N;
            const int64_t num = (int64_t)qlen * (G.A.D.sc_mch - worst) - G.A.D.qo;
            const uint64_t keep = qlen + (num > 0 ? (uint64_t)(num / G.A.D.e) : 0ull);
            if (keep < tlen) { flag |= (int)(tlen << 16); tlen = keep; }
        }
        if (WRITE) {
            moni_dp_task_t t;
            t.q_off = q_off; t.t_off = t_off; t.qlen = (int32_t)qlen; t.tlen = (int32_t)tlen; t.flag = flag; t.reserved = DP_Q_READS | DP_T_TEXT | qmode | tmode;
            PL.tasks[task0 + nt] = t;
        }
        ++nt;
    };
    if (CLASSIFY) { C.has_lc = C.has_rc = C.n_gap_tasks = 0; C.overlap = 0; }
    const af_anchor_t first = AN[0], last = AN[ch.cnt - 1];
    const uint32_t strand = C.strand;
#define qseg(a, len, reversed, q_off, qmode) af_qseg(off, m, strand, (a), (len), (reversed), (q_off), (qmode))
    const uint64_t lcs_len = first.idx, rcs_occ = (uint64_t)last.idx + last.len, rcs_len = m - rcs_occ;
    const uint64_t mem_pos = first.occ;
    if (lcs_len > 0) {
        const uint64_t lc_occ = mem_pos > ext_len ? mem_pos - ext_len : 0;
        const uint64_t lc_len = mem_pos > ext_len ? ext_len : ext_len - mem_pos;     // sic (aligner_ksw2.hpp:2796)
        uint64_t q_off; int qmode;
        qseg(0, lcs_len, true, q_off, qmode);
        add(q_off, lcs_len, qmode, lc_len ? lc_occ + lc_len - 1 : 0, lc_len, DP_T_REV, DP_EZ_EXTZ_ONLY | DP_EZ_RIGHT);
        if (CLASSIFY) C.has_lc = 1;
    }
    if (rcs_len > 0) {
        const uint64_t rc_occ = last.occ + last.len;
        const uint64_t rc_len = rc_occ < n_text - ext_len ? ext_len : n_text - rc_occ;
        uint64_t q_off; int qmode;
        qseg(rcs_occ, rcs_len, false, q_off, qmode);
        add(q_off, rcs_len, qmode, rc_occ, rc_len, 0, DP_EZ_EXTZ_ONLY | DP_EZ_RIGHT);
        if (CLASSIFY) C.has_rc = 1;
    }
    uint64_t last_ref = mem_pos + first.len, last_seq = (uint64_t)first.idx + first.len;
    if (CLASSIFY) {
        for (uint32_t k = 1; k < ch.cnt; ++k) {                  // overlapping anchors: the global realignment of the whole read (aligner_ksw2.hpp:2888-2900, 2984-2996)
            const af_anchor_t ak = AN[k];
            if (last_ref > ak.occ || last_seq > ak.idx) C.overlap = 1;
            last_ref = ak.occ + ak.len; last_seq = (uint64_t)ak.idx + ak.len;
        }
        if (C.overlap && m > AF_QCAP) bad = true;
        last_ref = mem_pos + first.len; last_seq = (uint64_t)first.idx + first.len;
    }
    for (uint32_t k = 1; k < ch.cnt && !C.overlap; ++k) {
        const af_anchor_t ak = AN[k], ap = AN[k - 1];
        af_anchor_t& GP = AN[k - 1];
        const uint64_t ref_occ = ak.occ, seq_occ = ak.idx;
        if (MODE == 1) {                                         // the gaps were classified by the counting pass
            if (ap.gap_kind == AF_GAP_TASK) {
                const uint64_t cc_occ = ap.occ + ap.len, cc_len = ref_occ - cc_occ;
                const uint64_t ccs_pos = (uint64_t)ap.idx + ap.len, ccs_len = seq_occ - ccs_pos;
                uint64_t q_off; int qmode;
                qseg(ccs_pos, ccs_len, false, q_off, qmode);
                add(q_off, ccs_len, qmode, cc_occ, cc_len, 0, DP_EZ_RIGHT);
            }
            continue;
        }
        if (last_ref == ref_occ) {
            if (last_seq < seq_occ) {                                              // pure insertion
                GP.gap_val = (int16_t)(seq_occ - last_seq); GP.gap_kind = AF_GAP_INS;       // < AF_MAX_READ
            }
        } else if (last_seq == seq_occ) {                                          // "deletion": l is computed as 0 (aligner_ksw2.hpp:2939)
            GP.gap_val = (int16_t)af_ins_score(P, 0); GP.gap_kind = AF_GAP_DEL0;
        } else {
            const uint64_t cc_occ = ap.occ + ap.len, cc_len = ref_occ - cc_occ;
            const uint64_t ccs_pos = (uint64_t)ap.idx + ap.len, ccs_len = seq_occ - ccs_pos;
            uint64_t q_off; int qmode;
            qseg(ccs_pos, ccs_len, false, q_off, qmode);
            bool closed = false;
            if (ccs_len == 1 && cc_len == 1) {       // one base against one base: the diagonal move wins unless both gaps beat it
                uint32_t qc = dp_nt4(G.A.D.reads[q_off]);
                if ((qmode & DP_Q_COMP) && qc < 4) qc = 3 - qc;
                const uint32_t tc = dp_nt4(cc_occ < n_text ? G.A.D.text[cc_occ] : 0u);
                if (qc < 4 && tc < 4) {
                    const int32_t z = tc == qc ? G.A.D.sc_mch : G.A.D.sc_mis;
                    const int32_t gap = dp_bound(0, G.A.D.qo, G.A.D.e) - G.A.D.qo - G.A.D.e;
                    if (z > gap) { GP.gap_val = (int16_t)z; GP.gap_kind = AF_GAP_1X1; closed = true; }
                }
            }
            if (!closed) {
                add(q_off, ccs_len, qmode, cc_occ, cc_len, 0, DP_EZ_RIGHT);
                GP.gap_kind = AF_GAP_TASK; C.n_gap_tasks++;
            }
        }
        last_ref = ref_occ + ak.len; last_seq = seq_occ + ak.len;
    }
#undef qseg
    return bad ? 0xFFFFFFFFu : nt;
}

// After the lanes have lifted every chain's leftmost anchor: the chain-selection loop ahead of its scores (aligner_ksw2.hpp:409-462) - which chains
// it scores - by lane 0 (a few comparisons per chain, no memory traffic), then one LANE per chain to score builds that chain's anchors and problems
// (af_build_cand: the one-base gaps read the read and the text - dependent HBM loads that one lane used to take one after the other for every chain).
// All lanes call it; the returned status is uniform: AF_ST_CAND, a fallback, or 0xFF (does not fit this instance: the next larger one takes the read).
template <class WT, class GT>
__device__ __forceinline__ uint32_t af_plan_cands(const af_args_t& G, WT& L, uint64_t off, uint32_t m, const GT g) {
    const ac_params_t& P = G.A.P;
    const int lane = g.lane;
    static_assert(WT::NC <= GT::W, "one lane per chain to score");
    const uint32_t n_chains = L.n_chains_sh;
    auto& PL = L.plan;
    // a capacity of THIS instance (chains to score, their anchors, their problems): the next larger instance takes the read; the largest hands it to align_kernel
    constexpr bool has_larger = (uint32_t)WT::NC < AF_MAX_CAND || (uint32_t)WT::NA < AF_PLAN_AN || (uint32_t)WT::NT < AF_MAX_TASKS_READ;
    if (lane == 0) {
        PL.n_chains = (uint16_t)n_chains; PL.n_cand = 0; PL.n_an = 0;
        L.n_tasks = 0;
        uint32_t st = AF_ST_CAND, why = AF_WHY_N;
        int64_t* diff = L.diff; uint32_t n_diff = 0, n_left = 0;
        for (uint32_t ci = 0; ci < n_chains && n_diff < P.check_k; ++ci) {
            const af_chain_t ch = L.chains[ci];
            { bool f = false; for (uint32_t q = 0; q < n_diff; ++q) f = f || diff[q] == (int64_t)ch.score; if (!f) diff[n_diff++] = ch.score; }
            if (P.left_mem_check) {                                  // check_left_MEM (aligner_ksw2.hpp:553-597)
                const uint64_t left_ref = L.left_ref[ci];
                bool seen = false;
                for (uint32_t k = 0; k < n_left; ++k) {
                    const uint64_t lr = L.left_ref[L.left_idx[k]];
                    const uint64_t d = lr > left_ref ? lr - left_ref : left_ref - lr;
                    if (d < P.region_dist && L.chains[L.left_idx[k]].score == ch.score) seen = true;
                }
                if (seen) continue;
                L.left_idx[n_left++] = (uint16_t)ci;
            }
            if (n_diff >= P.check_k) continue;                       // not scored; the loop condition ends the loop
            if (PL.n_cand >= (uint32_t)WT::NC) { st = 0xFFu; why = AF_WHY_CANDS; break; }
            if (ch.cnt > 255u || PL.n_an + ch.cnt > (uint32_t)WT::NA) { st = 0xFFu; why = AF_WHY_CHAIN_LEN; break; }
            af_cand_t& C = PL.cand[PL.n_cand];
            C.chain_score = ch.score; C.chain_idx = (uint16_t)ci; C.n_an = (uint8_t)ch.cnt; C.task0 = 0; C.has_lc = C.has_rc = C.n_gap_tasks = 0; C.overlap = 0; C.score = 0; C.gtask = 0;
            C.strand = 0; C.an0 = (uint16_t)PL.n_an; C.pad = 0; C.pad2 = 0;
            PL.n_an += ch.cnt;
            PL.n_cand++;
        }
        if (st == 0xFFu && !has_larger) st = AF_FALLBACK(G, why);
        L.status_sh = st;
    }
    __syncthreads();
    uint32_t status = L.status_sh;
    if (status != AF_ST_CAND) return status;
    const uint32_t n_cand = PL.n_cand;
    // ---- the problems of every chain to score: lane c takes chain c; count, prefix sum, write ----
    uint32_t nt = 0;
    if ((uint32_t)lane < n_cand) { af_cand_anchors(L, PL.cand[lane], 3u, 0u); nt = af_build_cand<false>(G, L, PL.cand[lane], off, m, 0u); }
    const bool bad = g.ballot(nt == 0xFFFFFFFFu) != 0ull;
    if (bad) { if (lane == 0) L.status_sh = AF_FALLBACK(G, AF_WHY_TASK_SIZE); __syncthreads(); return L.status_sh; }
    uint32_t incl = nt;
    for (int o = 1; o < AF_MAX_CAND; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
    const uint32_t total = (uint32_t)g.shfl((int)incl, (int)(n_cand ? n_cand - 1 : 0));
    if (n_cand && total > (uint32_t)WT::NT) {
        if (has_larger) return 0xFFu;
        if (lane == 0) L.status_sh = AF_FALLBACK(G, AF_WHY_CAPACITY);
        __syncthreads();
        return L.status_sh;
    }
    if ((uint32_t)lane < n_cand) {
        PL.cand[lane].task0 = incl - nt;
        af_build_cand<true>(G, L, PL.cand[lane], off, m, incl - nt);
    }
    if (lane == 0) L.n_tasks = n_cand ? total : 0u;
    __syncthreads();
    return AF_ST_CAND;
}

// classify_kernel: which LDS instance of chain_plan_kernel a read needs, from its seeds alone (lane = read: the frequency filter's counts of kept seeds
// and of their occurrences = anchors).  The instances take their reads from three lists; before, every read went through the small instance, which found
// out after loading the seeds - and appended to the next list with one atomic per read on one address (~60 M/s: 8 ms per 1 M reads on a pangenome of 20
// haplotypes, where most reads have more than 96 anchors).  Here a wave appends its reads with one atomic per list.
__global__ void __launch_bounds__(256) classify_kernel(const af_args_t G) {
    const ak_args_t& A = G.A;
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t level = 4;                                      // 0 small, 1 large, 2 largest, 3 align_kernel, 4 no read
    uint32_t m = 0;
    if (r_in < A.n_reads) {
        const uint64_t r = A.read_lo + r_in;
        m = (uint32_t)(A.offs[r + 1] - A.offs[r]);
        const uint64_t a = A.read_mem_off[r], b = A.read_mem_off[r + 1];
        if (m >= AF_MAX_READ || (b - a) > (uint64_t)af_wave_huge_t::MM) level = 3;
        else {
            unsigned long long total = 0;
            for (uint64_t k = a; k < b; ++k) total += A.mems[k].occ_cnt;
            uint32_t n_mems = 0; unsigned long long n_anch = 0;
            for (uint64_t k = a; k < b; ++k) {
                const uint32_t oc = A.mems[k].occ_cnt;
                bool keep = true;
                if (A.P.filter_freq) { const double fr = static_cast<double>(oc) / (double)(size_t)total; if (fr > A.P.freq_thr) keep = false; }      // seed_freq_filter
                if (keep) { ++n_mems; n_anch += oc; }
            }
            level = (n_mems <= G.l0_mm && n_anch <= (unsigned long long)G.l0_ma) ? 0u
                  : (n_mems <= (uint32_t)af_wave_t::MM && n_anch <= (unsigned long long)af_wave_t::MA) ? 1u
                  : (n_mems <= (uint32_t)af_wave_huge_t::MM && n_anch <= (unsigned long long)af_wave_huge_t::MA) ? 2u : 3u;
        }
    }
#pragma unroll
    for (uint32_t lv = 0; lv < 3; ++lv) {
        const unsigned long long bal = __ballot(level == lv);
        if (!bal) continue;
        uint32_t base = 0;
        const int lead = __ffsll((long long)bal) - 1;
        if (lane == lead) base = atomicAdd(&G.ctr[lv == 0 ? AFC_L0 : lv == 1 ? AFC_BIG : AFC_HUGE], (uint32_t)__popcll(bal));
        base = (uint32_t)__shfl((int)base, lead);
        if (level == lv) (lv == 0 ? G.list0 : lv == 1 ? G.big_list : G.huge_list)[base + (uint32_t)__popcll(bal & lt_mask)] = (uint32_t)r_in;
    }
    if (level == 3) {                                        // beyond every instance: align_kernel
        af_plan_t& PL = G.plans[r_in];
        PL.status = AF_ST_FALLBACK; PL.n_cand = 0; PL.n_chains = 0; PL.final_cand = 0; PL.n_alt = 0; PL.pad = 0; PL.score2 = 0; PL.ref_pos = PL.ref_len = 0; PL.tb0 = 0; PL.pad2 = 0;
        PL.min_score = A.min_score_of_len[m <= A.max_len ? m : A.max_len];
        G.ntasks[r_in] = 0;
        G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)(A.read_lo + r_in);
        atomicAdd(&G.ctr[AFC_WHY + (m >= AF_MAX_READ ? AF_WHY_LONG : AF_WHY_ANCHORS)], 1u);
    }
}

// One read of chain_plan_kernel once its seeds and anchors are in LDS: chains (af_chain), check_left_MEM's lifts, the selection loop and the plan
// (af_plan_cands), the hand-over lists, the plan and the tasks to HBM.  All lanes of the read's group call it.
template <class WT, int LEVEL, class GT>
__device__ __forceinline__ void af_plan_read(const af_args_t& G, WT& L, const GT g, uint32_t r_in, uint64_t off, uint32_t m, uint32_t na, float avg, bool fallback, bool too_big) {
    constexpr uint32_t GW = GT::W;
    const int lane = g.lane;
    const ak_args_t& A = G.A;
    const uint64_t r = A.read_lo + r_in;
    uint32_t status = AF_ST_UNALIGNED;
#if defined(AF_PROFILE) || defined(AF_CUTS)
    if (G.dbg & 4) na = 0;                                   // timing experiments (results are wrong): stop after the anchors ...
#endif
    if (!fallback && !too_big && na > 0) {
        status = af_chain(G, L, na, avg, g);
        status = (uint32_t)g.shfl((int)status, 0);
        __syncthreads();
#if defined(AF_PROFILE) || defined(AF_CUTS)
        if ((G.dbg & 8) && status == AF_ST_CAND) status = AF_ST_UNALIGNED;      // ... after the chains ...
#endif
        if (status == 0xFFu) too_big = true;
        else if (status == AF_ST_CAND) {
            // check_left_MEM's coordinate of every chain: index(lift(leftmost anchor)).second + 1 (aligner_ksw2.hpp:565-576)
            for (uint32_t ci = lane; ci < L.n_chains_sh; ci += GW) {
                const af_chain_t ch = L.chains[ci];
                const uint64_t aw = L.anch[L.pool[ch.off + ch.cnt - 1]];
                const af_mem_t ml = L.mem[aw >> 40];
                L.left_ref[ci] = ac_seq_off(A.P, ac_lift(A.P, AF_X(aw) - ml.len + 1)) + 1;
            }
            __syncthreads();
#if defined(AF_PROFILE) || defined(AF_CUTS)
            if (G.dbg & 16) status = AF_ST_UNALIGNED; else      // ... after the lifts
#endif
            if (LEVEL == 0 && (uint32_t)WT::MC <= AF_CTAB && (uint32_t)WT::MA <= AF_PLAN_AN && G.ctab != nullptr) {
                // The selection loop and the plan of the DP problems are one-lane work (a third of this kernel's instructions at one read per wavefront): the chains
                // go to HBM - a table entry per chain and ALL anchors, chain by chain and left to right, straight into the plan's anchor array - and plan_kernel
                // does that work with one LANE per read (64 reads per wavefront).
                const uint32_t n_chains = L.n_chains_sh;
                af_ctab_t* const CT = G.ctab + (size_t)r_in * AF_CTAB;
                af_anchor_t* const AN = G.plans[r_in].an;
                uint32_t an0 = 0;                                  // exclusive prefix sum of the chains' anchor counts over the lanes
                for (uint32_t c0 = 0; c0 < n_chains; c0 += GW) {
                    const uint32_t ci = c0 + (uint32_t)lane;
                    const uint32_t cnt = ci < n_chains ? L.chains[ci].cnt : 0u;
                    uint32_t incl = cnt;
                    for (int o = 1; o < (int)GW; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += x; }
                    if (ci < n_chains) {
                        const af_chain_t ch = L.chains[ci];
                        const uint32_t at = an0 + incl - cnt;
                        uint32_t strand = 0;
                        for (uint32_t k = 0; k < ch.cnt; ++k) {          // stored right to left (chain.hpp:166-200); fill_chain wants left to right
                            const uint64_t aw = L.anch[L.pool[ch.off + ch.cnt - 1 - k]];
                            const af_mem_t mk = L.mem[aw >> 40];
                            af_anchor_t X; X.occ = AF_X(aw) - mk.len + 1; X.len = mk.len; X.idx = mk.idx; X.gap_val = 0; X.gap_kind = AF_GAP_NONE; X.pad = 0;
                            AN[at + k] = X;
                            if (k == 0) strand = (mk.mate & 2) ? 1u : 0u;
                        }
                        af_ctab_t E; E.score = ch.score; E.an0 = (uint16_t)at; E.cnt = (uint8_t)ch.cnt; E.strand = (uint8_t)strand; E.left_ref = L.left_ref[ci];
                        CT[ci] = E;
                    }
                    an0 += (uint32_t)g.shfl((int)incl, (int)GW - 1);
                }
                if (lane == 0) {          // the plan's header so far; plan_kernel completes it
                    af_plan_t& PH = G.plans[r_in];
                    PH.status = AF_ST_CHAINS; PH.n_cand = 0; PH.n_chains = (uint16_t)n_chains; PH.final_cand = 0; PH.n_alt = 0; PH.pad = 0; PH.score2 = 0; PH.ref_pos = PH.ref_len = 0; PH.tb0 = 0; PH.pad2 = 0;
                    PH.min_score = A.min_score_of_len[m <= A.max_len ? m : A.max_len];
                    G.ntasks[r_in] = 0;
                }
                __syncthreads();
                return;
            }
            status = af_plan_cands(G, L, off, m, g);
            __syncthreads();
            if (status == 0xFFu) too_big = true;              // the plan does not fit this instance
        }
    }
    __syncthreads();
    if (too_big) {
        if (LEVEL == 0) { if (lane == 0) G.big_list[atomicAdd(&G.ctr[AFC_BIG], 1u)] = r_in; return; }      // the next larger instance takes it
        if (LEVEL == 1) { if (lane == 0) G.huge_list[atomicAdd(&G.ctr[AFC_HUGE], 1u)] = r_in; return; }
        fallback = true;
    }
    if (fallback) { status = AF_ST_FALLBACK; if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + (m >= AF_MAX_READ ? AF_WHY_LONG : AF_WHY_ANCHORS)], 1u); }
    // ---- the read's tasks go to the batch's list and the bins of their tile / query length; the plan goes to HBM ----
    auto& PL = L.plan;
    const uint32_t nt = status == AF_ST_CAND ? L.n_tasks : 0u;
    const uint32_t t0 = r_in * AF_MAX_TASKS_READ;                 // the read's own slots (bin_tasks_kernel queues them)
    if (status == AF_ST_CAND) {
        for (uint32_t k = lane; k < nt; k += GW) G.tasks[t0 + k] = PL.tasks[k];
        if ((uint32_t)lane < PL.n_cand) PL.cand[lane].task0 += t0;
    }
    if (lane == 0) G.ntasks[r_in] = (uint8_t)nt;
    if (lane == 0) {
        if (status != AF_ST_CAND) { PL.n_cand = 0; PL.n_chains = 0; }
        PL.status = (uint8_t)status; PL.final_cand = 0; PL.n_alt = 0; PL.pad = 0; PL.score2 = 0; PL.ref_pos = PL.ref_len = 0; PL.tb0 = 0; PL.pad2 = 0;
        PL.min_score = A.min_score_of_len[m <= A.max_len ? m : A.max_len];
        if (status == AF_ST_FALLBACK) G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)r;
    }
    __syncthreads();
    {   // plan: only the parts in use (header + chains to score; their anchors)
        const uint32_t words = (uint32_t)((offsetof(af_plan_t, cand) + (status == AF_ST_CAND ? PL.n_cand : 0u) * sizeof(af_cand_t)) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&PL);
        uint32_t* dst = reinterpret_cast<uint32_t*>(G.plans + r_in);
        for (uint32_t w = lane; w < words; w += GW) dst[w] = src[w];
        const uint32_t awords = status == AF_ST_CAND ? PL.n_an * (uint32_t)(sizeof(af_anchor_t) / 4) : 0u;
        const uint32_t* asrc = reinterpret_cast<const uint32_t*>(PL.an);
        uint32_t* adst = reinterpret_cast<uint32_t*>(G.plans[r_in].an);
        for (uint32_t w = lane; w < awords; w += GW) adst[w] = asrc[w];
    }
    __syncthreads();
}

// WT: the LDS instance (capacities).  LEVEL 0 / 1 / 2: the reads of list0 / big_list / huge_list (classify_kernel); a read whose chains or plan overflow
// the instance it was given goes to the next list, from the largest instance to align_kernel.  GW: the lanes of one read (af_grp_t): 64 / GW reads per
// wavefront side by side, each in its own copy of WT.
// (Measured and not kept, profiles/r04f: classify_kernel gathering a read's seeds and anchors into a slot that the LEVEL-0 instance asks for one read ahead - the
// kernel is bound by VALU issue, not by those loads: 9.36 against 9.2 ms per 1 M reads, and 1.5 ms more in classify_kernel.)
template <class WT, int LEVEL, int OCC = (LEVEL == 0 ? 6 : LEVEL == 1 ? 3 : 1), int GW = 64>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) chain_plan_kernel(const af_args_t G) {
    constexpr uint32_t NG = 64 / GW;          // reads per wavefront
    __shared__ WT Ls[NG];                     // one wavefront per workgroup: __syncthreads() orders the wave's own LDS traffic
    typedef af_grp_t<GW> GT;
    const GT g;
    WT& L = Ls[NG > 1 ? g.id() : 0];
    const int lane = g.lane;
    const ak_args_t& A = G.A;
    constexpr bool BIG = LEVEL > 0;
    const uint32_t n_work = G.ctr[LEVEL == 0 ? AFC_L0 : LEVEL == 1 ? AFC_BIG : AFC_HUGE];
    const uint32_t* const work_list = LEVEL == 0 ? G.list0 : LEVEL == 1 ? G.big_list : G.huge_list;
    constexpr uint32_t GRAB = BIG ? 1u : NG > 1 ? 2u : 8u;          // reads a group takes per visit of the wavefront to the shared cursor
    uint32_t w_base = 0, w_k = GRAB;          // wave-uniform
    while (true) {
        if (w_k >= GRAB) {
            if (threadIdx.x == 0) w_base = atomicAdd(&G.ctr[LEVEL == 0 ? AFC_READ_CUR : LEVEL == 1 ? AFC_BIG_CUR : AFC_HUGE_CUR], GRAB * NG);
            w_base = (uint32_t)__shfl((int)w_base, 0);
            w_k = 0;
        }
        if (w_base + w_k * NG >= n_work) break;          // (wave-uniform: the groups leave together)
        const uint32_t w_in = w_base + w_k * NG + (NG > 1 ? (uint32_t)g.id() : 0u);
        ++w_k;
        if (w_in >= n_work) continue;                    // the list's last reads: fewer than the wavefront has groups
        const uint32_t r_in = work_list[w_in];
        const uint64_t r = A.read_lo + r_in;
        const uint64_t off = A.offs[r];
        const uint32_t m = (uint32_t)(A.offs[r + 1] - off);
        const uint64_t a = A.read_mem_off[r], b = A.read_mem_off[r + 1];
        if (lane == 0) L.n_tasks = 0;
        bool fallback = m >= AF_MAX_READ || (b - a) > 4 * AF_MAX_MEMS;      // not for any instance
        bool too_big = false;                                              // not for this instance
        uint32_t n_mems = 0, na = 0;
        float avg = 0.f;
        if (!fallback && b > a) {
            // ---- seed_freq_filter (aligner_ksw2.hpp:1905-1933): lanes over the read's seeds ----
            unsigned long long total = 0;
            for (uint64_t k = a + lane; k < b; k += GW) total += A.mems[k].occ_cnt;
            for (int o = GW / 2; o > 0; o >>= 1) total += __shfl_xor(total, o);
            unsigned long long tot_len = 0, n_anch = 0;
            for (uint64_t k0 = a; k0 < b; k0 += GW) {
                const uint64_t k = k0 + lane;
                bool keep = false;
                moni_mem_t gm;
                if (k < b) {
                    gm = A.mems[k];
                    keep = true;
                    if (A.P.filter_freq) { const double fr = static_cast<double>(gm.occ_cnt) / (double)(size_t)total; if (fr > A.P.freq_thr) keep = false; }
                }
                const unsigned long long bal = g.ballot(keep);
                const uint32_t at = n_mems + (uint32_t)__popcll(bal & g.lt_mask());
                if (keep && at < (uint32_t)WT::MM) {
                    af_mem_t x; x.occ_off = gm.occ_off; x.nocc = gm.occ_cnt; x.len = (uint16_t)gm.len; x.idx = (uint16_t)gm.idx; x.rpos = (uint16_t)gm.rpos; x.mate = (uint8_t)gm.mate; x.pad = 0;
                    L.mem[at] = x;
                }
                if (keep) { tot_len += (unsigned long long)gm.len * gm.occ_cnt; n_anch += gm.occ_cnt; }
                n_mems += (uint32_t)__popcll(bal);
            }
            for (int o = GW / 2; o > 0; o >>= 1) { tot_len += __shfl_xor(tot_len, o); n_anch += __shfl_xor(n_anch, o); }
            if (n_mems > (uint32_t)WT::MM || n_anch > (unsigned long long)WT::MA) too_big = true;
            na = (uint32_t)n_anch;
            if (!too_big && na > 0) {
                avg = (float)(size_t)tot_len / (size_t)n_anch;
                __syncthreads();
                // ---- populate_anchors (chain.hpp:83-95): mem by mem, occurrence by occurrence ----
                if (NG > 1) {          // few lanes: every lane finds the seed of its anchors (a seed has about as many occurrences as the index has sequences)
                    for (uint32_t x = lane; x < na; x += GW) {
                        uint32_t i = 0, base = 0;
                        while (base + L.mem[i].nocc <= x) { base += L.mem[i].nocc; ++i; }
                        const af_mem_t mi = L.mem[i];
                        L.anch[x] = (A.occs[mi.occ_off + (x - base)] + mi.len - 1) | ((uint64_t)i << 40);
                    }
                } else {
                    uint32_t base = 0;
                    for (uint32_t i = 0; i < n_mems; ++i) {
                        const af_mem_t mi = L.mem[i];
                        for (uint32_t j = lane; j < mi.nocc; j += GW) L.anch[base + j] = (A.occs[mi.occ_off + j] + mi.len - 1) | ((uint64_t)i << 40);
                        base += mi.nocc;
                    }
                }
                for (uint32_t i = lane; i < na; i += GW) { L.f[i] = 0; L.msc[i] = 0; L.p[i] = 0; L.t[i] = 0; }
            }
        }
        __syncthreads();
        af_plan_read<WT, LEVEL>(G, L, g, r_in, off, m, na, avg, fallback, too_big);
    }
}

// plan_kernel: one LANE per read whose chains chain_plan_kernel's LEVEL-0 instance has left in HBM (status AF_ST_CHAINS): the chain-selection loop ahead of its
// scores (aligner_ksw2.hpp:409-462: which chains get scored - af_plan_cands' loop, with the capacities of the largest instance), then for every chain to score
// its gaps and DP problems (af_build_cand, one pass) straight into the read's task slots and plan.  The lists the loop keeps (distinct scores, the chains
// check_left_MEM has recorded) live in the lane's own LDS columns.
struct af_plan_view_t { af_anchor_t* an; moni_dp_task_t* tasks; };
struct af_view_t { af_plan_view_t plan; };
__global__ void __launch_bounds__(256) plan_kernel(const af_args_t G) {
    __shared__ int32_t s_diff[8][256];
    __shared__ uint8_t s_left[AF_CTAB][256];          // (the recorded chains' coordinates are read again from the table: 16-byte entries of the lane's own 768 bytes)
    const int lane = threadIdx.x;
    const ak_args_t& A = G.A;
    const ac_params_t& P = A.P;
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r_in >= A.n_reads) return;
    af_plan_t& PL = G.plans[r_in];
    if (PL.status != AF_ST_CHAINS) return;
    const uint64_t r = A.read_lo + r_in;
    const uint64_t off = A.offs[r];
    const uint32_t m = (uint32_t)(A.offs[r + 1] - off);
    const uint32_t n_chains = PL.n_chains;
    const af_ctab_t* const CT = G.ctab + (size_t)r_in * AF_CTAB;
    const uint32_t t0 = (uint32_t)r_in * AF_MAX_TASKS_READ;
    af_view_t V; V.plan.an = PL.an; V.plan.tasks = G.tasks + t0;
    uint32_t status = AF_ST_CAND, why = AF_WHY_N;
    uint32_t n_diff = 0, n_left = 0, n_cand = 0, n_tasks = 0;
    const uint32_t check_k = P.check_k < 8u ? P.check_k : 8u;          // (the loop ends at check_k distinct scores; the lane's list holds 8)
    if (P.check_k > 8u) { status = AF_ST_FALLBACK; why = AF_WHY_CAPACITY; }
    for (uint32_t ci = 0; ci < n_chains && n_diff < check_k && status == AF_ST_CAND; ++ci) {
        const af_ctab_t ch = CT[ci];
        { bool f = false; for (uint32_t q = 0; q < n_diff; ++q) f = f || s_diff[q][lane] == ch.score; if (!f) s_diff[n_diff++][lane] = ch.score; }
        if (P.left_mem_check) {                                  // check_left_MEM (aligner_ksw2.hpp:553-597)
            bool seen = false;
            for (uint32_t k = 0; k < n_left; ++k) {
                const af_ctab_t o = CT[s_left[k][lane]];
                const uint64_t d = o.left_ref > ch.left_ref ? o.left_ref - ch.left_ref : ch.left_ref - o.left_ref;
                if (d < P.region_dist && o.score == ch.score) seen = true;
            }
            if (seen) continue;
            s_left[n_left][lane] = (uint8_t)ci; ++n_left;
        }
        if (n_diff >= check_k) continue;                       // not scored; the loop condition ends the loop
        if (n_cand >= AF_MAX_CAND) { status = AF_ST_FALLBACK; why = AF_WHY_CANDS; break; }
        af_cand_t C;
        C.chain_score = ch.score; C.chain_idx = (uint16_t)ci; C.n_an = ch.cnt; C.task0 = t0 + n_tasks; C.has_lc = C.has_rc = C.n_gap_tasks = 0; C.overlap = 0; C.score = 0; C.gtask = 0;
        C.strand = ch.strand; C.an0 = ch.an0; C.pad = 0; C.pad2 = 0;
        // its problems behind those of the chains before it; at most AF_MAX_TASKS_READ in all (a chain has at most 2 + n_an - 1)
        if (n_tasks + 1u + ch.cnt > AF_MAX_TASKS_READ) { status = AF_ST_FALLBACK; why = AF_WHY_CAPACITY; break; }
        const uint32_t nt = af_build_cand<2>(G, V, C, off, m, n_tasks);
        if (nt == 0xFFFFFFFFu) { status = AF_ST_FALLBACK; why = AF_WHY_TASK_SIZE; break; }
        n_tasks += nt;
        PL.cand[n_cand++] = C;
    }
    if (status == AF_ST_FALLBACK) {
        atomicAdd(&G.ctr[AFC_WHY + why], 1u);
        G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)r;
        PL.status = AF_ST_FALLBACK; PL.n_cand = 0; PL.n_chains = 0;
        G.ntasks[r_in] = 0;
        return;
    }
    PL.status = AF_ST_CAND; PL.n_cand = (uint8_t)n_cand;
    G.ntasks[r_in] = (uint8_t)n_tasks;
}

// bin_tasks_kernel: the reads' DP problems go to the queues of their tile / query-length bins.  A block takes AF_BT_READS reads (rounds of 4 reads x 64
// task slots, one thread per slot), counts per bin in LDS and bumps every bin's global counter once: a few thousand atomics per address and launch where
// one per task was the bound of chain_plan_kernel (and one per bin and 4 reads, with 35 bins instead of 18, was 2.9 ms per 1 M reads: profiles/r04e).
#define AF_BT_READS 32u
#define AF_T_WILDQ 0x4000          // in a global problem's `reserved`: its read holds a wildcard base (global_task_kernel -> global_band_kernel)
__device__ __forceinline__ void af_global_band(const dp_launch_t& D, const moni_dp_task_t& T, int& dmin, int& dmax, bool* saw_wild = nullptr);
__device__ __forceinline__ void af_ext_band(const dp_launch_t& D, const moni_dp_task_t& T, int& dmin, int& dmax);
__global__ void __launch_bounds__(256) bin_tasks_kernel(const af_args_t G) {
    __shared__ uint32_t cnt[AF_NBIN + 1], base[AF_NBIN + 1], ovf[AF_BT_READS];
    static_assert(AF_MAX_TASKS_READ == 64, "bin_tasks_kernel: 4 reads x 64 slots per round");
    constexpr uint32_t ROUNDS = AF_BT_READS / 4;
    const uint32_t tid = threadIdx.x;
    const uint64_t n_reads = G.A.n_reads;
    if (blockIdx.x == 0 && tid == 0) G.ctr[AFC_TASKS] = (uint32_t)n_reads * AF_MAX_TASKS_READ;      // the global problems (global_task_kernel) come after the slots
    if (tid <= AF_NBIN) cnt[tid] = 0;
    if (tid < AF_BT_READS) ovf[tid] = 0;
    __syncthreads();
    const uint32_t k = tid & 63u;
    uint32_t bin[ROUNDS], local[ROUNDS];
    bool valid[ROUNDS];
#pragma unroll
    for (uint32_t q = 0; q < ROUNDS; ++q) {
        const uint64_t r_in = (uint64_t)blockIdx.x * AF_BT_READS + 4 * q + (tid >> 6);
        valid[q] = r_in < n_reads && k < (uint32_t)G.ntasks[r_in];
        bin[q] = 0; local[q] = 0;
        if (valid[q]) {
            const uint32_t id = (uint32_t)r_in * AF_MAX_TASKS_READ + k;
            const moni_dp_task_t t = G.tasks[id];
            bin[q] = af_bin_of(t.qlen, t.tlen);
            // a wildcard base may lie in the problem (the read holds one, or the target touches a block of the text that does): the WILDC instance's queues
            {
                uint64_t rd = G.A.read_lo + r_in;
                if (G.pe) { rd *= 2; if (t.q_off >= G.A.offs[rd + 1]) ++rd; }
                if (!G.pflag || G.pflag[2 * rd + ((t.reserved & DP_Q_COMP) ? 1u : 0u)] || af_target_exc(G, t)) bin[q] = AF_BIN_WILD + af_large_bin(t.qlen);
            }
            // a problem of the large tile - an extension, or a gap fill (a small global problem) - goes to a provisional list: band_tasks_kernel, one lane per entry,
            // bounds its diagonals and queues it for dp_band_kernel or for the tile (MONI_AF_DBG=131072: the tile kernels take all)
            if (bin[q] < AF_BIN_SMALL && !(G.dbg & 0x20000u)) bin[q] = AF_BIN_PROV;
            local[q] = atomicAdd(&cnt[bin[q]], 1u);
        }
    }
    __syncthreads();
    if (tid <= AF_NBIN && cnt[tid]) base[tid] = atomicAdd(&G.ctr[AFC_BINS + tid], cnt[tid]);
    if (tid == 0) { uint32_t n = 0; for (uint32_t b = 0; b <= AF_NBIN; ++b) n += cnt[b]; if (n) atomicAdd(&G.ctr[AFC_NT], n); }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < ROUNDS; ++q) if (valid[q]) {
        const uint32_t id = (uint32_t)((uint64_t)blockIdx.x * AF_BT_READS + 4 * q + (tid >> 6)) * AF_MAX_TASKS_READ + k;
        const uint32_t at = base[bin[q]] + local[q];
        if (at < G.bin_cap) { G.bin_q[(size_t)bin[q] * G.bin_cap + at] = id; G.task_pos[id] = at | (bin[q] << AF_POS_BITS); }
        else ovf[4 * q + (tid >> 6)] = 1;                      // the queue is full: the read goes to align_kernel
    }
    __syncthreads();
    if (tid < AF_BT_READS && ovf[tid]) {
        const uint64_t rr = (uint64_t)blockIdx.x * AF_BT_READS + tid;
        G.plans[rr].status = AF_ST_FALLBACK;
        G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)(G.A.read_lo + rr);
        atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CAPACITY], 1u);
    }
}

// 128-task chunks of a group of bins with the offsets of their direction bits (one wavefront: a lane per bin for the sizes, a prefix
// sum, then the lanes write the few thousand chunk records side by side).  Runs once for the extension / gap problems (groups large
// and small) and once more, after global_task_kernel, for the global problems.  Longest queries first: the persistent DP waves take
// chunks in this order, the short ones fill the tail.
__global__ void __launch_bounds__(64) af_chunk_kernel(const af_args_t G, const uint32_t first_group, const uint32_t last_group) {
    __shared__ uint32_t s_cnt[32], s_cbase[33], s_qhi[32], s_bin[32];
    __shared__ uint64_t s_dbase[32], s_bytes[32];
    const int lane = threadIdx.x;
    if (blockIdx.x != 0) return;
    uint32_t nc = 0;
    for (uint32_t g = 0; g < first_group; ++g) nc += G.ctr[AFC_NCHUNKS + g];
    uint64_t doff;
    memcpy(&doff, &G.ctr[AFC_DIROFF], 8);
    for (uint32_t grp = first_group; grp <= last_group; ++grp) {
        const uint32_t b0 = af_grp_b0(grp), b1 = af_grp_b1(grp);
        const uint32_t nb = b1 - b0;                     // <= 16
        const uint32_t tb = af_grp_tb(grp), np = af_grp_np(grp);
        uint32_t cnt = 0, nch = 0, qhi = 0, bin = 0; uint64_t bytes = 0;
        if ((uint32_t)lane < nb) {
            bin = b1 - 1 - (uint32_t)lane;
            cnt = G.ctr[AFC_BINS + bin] < G.bin_cap ? G.ctr[AFC_BINS + bin] : G.bin_cap;      // (a bin beyond its queue: bin_tasks_kernel sent the reads to align_kernel)
            nch = (cnt + 127) >> 7;
            qhi = af_bin_qhi(bin);
            bytes = (uint64_t)np * qhi * tb * 64;          // of one chunk: half a byte per cell
        }
        uint32_t cb = nch; uint64_t db = (uint64_t)nch * bytes;          // inclusive prefix sums over the lanes
        for (int o = 1; o < 32; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)cb, o); const uint64_t y = __shfl_up(db, o); if (lane >= o) { cb += x; db += y; } }
        if ((uint32_t)lane < nb) { s_cnt[lane] = cnt; s_qhi[lane] = qhi; s_bin[lane] = bin; s_bytes[lane] = bytes; s_cbase[lane] = cb - nch; s_dbase[lane] = db - (uint64_t)nch * bytes; }
        uint32_t total = (uint32_t)__shfl((int)cb, (int)nb - 1);
        const uint64_t total_bytes = __shfl(db, (int)nb - 1);
        if (lane == 0) s_cbase[nb] = total;
        __syncthreads();
        if (nc + total > G.chunk_cap) total = G.chunk_cap > nc ? G.chunk_cap - nc : 0;      // (cannot happen: the cap counts a chunk per 64 queue entries)
        for (uint32_t c = lane; c < total; c += 64) {
            uint32_t l = 0;
            while (l + 1 < nb && s_cbase[l + 1] <= c) ++l;
            const uint32_t j = c - s_cbase[l];
            af_chunk_t ch; ch.bin = s_bin[l]; ch.start = j * 128u; ch.n = s_cnt[l] - j * 128u < 128u ? s_cnt[l] - j * 128u : 128u; ch.qhi = s_qhi[l];
            const uint64_t at = doff + s_dbase[l] + (uint64_t)j * s_bytes[l];
            if (at + s_bytes[l] > G.dirs_cap) { ch.dir_off = ~0ull; G.ctr[AFC_DIRS_OVF] = 1; } else ch.dir_off = at;
            G.chunks[nc + c] = ch;
        }
        if (lane == 0) { G.ctr[AFC_NCHUNKS + grp] = total; G.ctr[AFC_WMODE + grp] = (grp == AF_GRP_WILD || grp == AF_GRP_GLOBAL || grp == AF_GRP_GWILD) && total <= G.wave_max ? 1u : 0u; }
        nc += total; doff += total_bytes;
        __syncthreads();
    }
    if (lane == 0) memcpy(&G.ctr[AFC_DIROFF], &doff, 8);
}

// A lane's operand bytes come as aligned 8-byte words (one request per eight bases instead of one per base: with 64 lanes on 64
// different cache lines the byte form is bound by request rate).  Sequence element k is at byte start + k (forward) or start - k
// (reverse); af_group returns elements 8g .. 8g+7 as one word, element 8g+u in byte u.  Words outside [0, limit) read as zero.
struct af_bytes_t { const uint8_t* base; int64_t start, limit; bool rev; uint64_t carry; int64_t carry_addr; };
__device__ __forceinline__ af_bytes_t af_bytes(const uint8_t* base, uint64_t start, uint64_t limit, bool rev) {
    af_bytes_t S; S.base = base; S.start = (int64_t)start; S.limit = (int64_t)limit; S.rev = rev; S.carry = 0; S.carry_addr = -8; return S;
}
__device__ __forceinline__ uint64_t af_group(af_bytes_t& S, int g) {
    const int64_t lo = S.rev ? S.start - 8 * g - 7 : S.start + 8 * g;
    const int64_t wa = lo & ~7ll;
    const int sh = (int)(lo & 7);
    uint64_t a = S.carry, b = S.carry;
    if (S.carry_addr != wa) a = (wa >= 0 && wa < S.limit) ? *reinterpret_cast<const uint64_t*>(S.base + wa) : 0ull;
    if (sh && S.carry_addr != wa + 8) b = (wa + 8 >= 0 && wa + 8 < S.limit) ? *reinterpret_cast<const uint64_t*>(S.base + wa + 8) : 0ull;
    uint64_t v = sh ? (a >> (8 * sh)) | (b << (64 - 8 * sh)) : a;
    if (S.rev) { S.carry = a; S.carry_addr = wa; v = __builtin_bswap64(v); }        // the next group ends where this one's first word ends
    else if (sh) { S.carry = b; S.carry_addr = wa + 8; }
    return v;
}

// ------------------------------------------------------------------------------------------------------------------------------
// dp_lane_kernel: ksw_extz2_sse (thirdparty/ksw2, absent; call sites aligner_ksw2.hpp:2812,2844,2965,2988,3015), one
// lane per problem.  Outer loop over the query (wave-uniform trip count), inner loop over a block of TB target rows fully unrolled:
// H(i, j-1) and F(i, j) of every row of the block in registers, E and the diagonal carried along the block.  NP > 1: targets longer
// than TB are taken block by block; (H, E) of a block's last row go through a per-wave buffer to the next block.
// ------------------------------------------------------------------------------------------------------------------------------
// Two problems per lane, their cells side by side in the 16-bit halves of every register (v_pk_* arithmetic: one instruction serves both):
// a chunk holds 128 problems, lane l takes the problems at l (low half) and 64 + l (high half) of the chunk.  Scores of reads up to
// AF_MAX_READ bases fit 16 bits with room for the "minus infinity" of E / F (AF_NEG16).  Per cell and problem four sign bits are kept
// instead of ksw2's direction byte (diagonal beats E, that maximum beats F, E does not continue, F does not continue: the negations of
// the tests of the right-aligned-gap rule), four rows per 16-bit half: half a byte per cell, decoded by traceback_kernel.
#define AF_NEG16 (-12000)
typedef short af_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short af_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ af_s2 af_as_s2(uint32_t x) { return __builtin_bit_cast(af_s2, x); }
__device__ __forceinline__ af_u2 af_as_u2(uint32_t x) { return __builtin_bit_cast(af_u2, x); }
__device__ __forceinline__ uint32_t af_pk(af_s2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t af_pku(af_u2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t af_pk_add(uint32_t a, uint32_t b) { return af_pk(af_as_s2(a) + af_as_s2(b)); }
__device__ __forceinline__ uint32_t af_pk_sub(uint32_t a, uint32_t b) { return af_pk(af_as_s2(a) - af_as_s2(b)); }
__device__ __forceinline__ uint32_t af_pk_max(uint32_t a, uint32_t b) { return af_pk(__builtin_elementwise_max(af_as_s2(a), af_as_s2(b))); }
__device__ __forceinline__ uint32_t af_pk_minu(uint32_t a, uint32_t b) { return af_pku(__builtin_elementwise_min(af_as_u2(a), af_as_u2(b))); }
__device__ __forceinline__ uint32_t af_pk_neg(uint32_t a) { return af_pku(af_as_u2(a) >> (af_u2)15); }      // 1 in a half that holds a negative value
__device__ __forceinline__ uint32_t af_pk2(int v) { return ((uint32_t)v & 0xFFFFu) * 0x10001u; }
__device__ __forceinline__ int af_lo16(uint32_t x) { return (int)(int16_t)(x & 0xFFFFu); }
__device__ __forceinline__ int af_hi16(uint32_t x) { return (int)(int16_t)(x >> 16); }

// WILDC: the instance for the problems that may hold a wildcard base (queues AF_BIN_WILD / AF_BIN_GWILD: bin_tasks_kernel and global_band_kernel send a problem
// there when its read has a byte outside A / C / G / T or its target touches such a block of the text).  It keeps two more planes (the rows and columns that
// hold a wildcard) and four more instructions per cell, which do not fit the registers beside a 52-row block - it runs with scratch; the plain instance,
// which takes everything else, is the code without them.
template <int TB, int QC, int NP, bool WILDC = false>
#ifndef AF_DP_OCC
#define AF_DP_OCC 2          // 52 rows x two problems + the cell temporaries fit 256 registers without scratch; at 3 waves the block spills
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TB > 64 ? 1 : AF_DP_OCC, TB > 64 ? 1 : AF_DP_OCC))) dp_lane_kernel(const af_args_t G, const uint32_t grp) {
    __shared__ uint8_t qs[QC][64];          // query codes of the lane's two problems: low one in bits 0-1 (bit 2: a wildcard), high one in bits 4-5 (bit 6)
    __shared__ uint32_t tns[WILDC ? (TB + 15) / 16 : 1][64];      // (WILDC) the block's target rows that hold a wildcard: row i of the low problem in bit i & 15, of the high one in bit 16 + (i & 15), of word i >> 4
    const int lane = threadIdx.x;
    const dp_launch_t& D = G.A.D;
    const uint32_t qo2 = af_pk2(D.qo), e2 = af_pk2(D.e), scM2 = af_pk2(D.sc_mch), scX2 = af_pk2(D.sc_mis), scN2 = af_pk2(D.sc_N); (void)scN2;
    const int32_t qo = D.qo, e = D.e;
    constexpr int NW = (TB + 15) / 16;
    uint64_t* __restrict__ bnd = NP > 1 ? G.bnd + (size_t)blockIdx.x * QC * 64 + lane : nullptr;
    uint32_t chunk0 = 0;
    for (uint32_t g = 0; g < grp; ++g) chunk0 += G.ctr[AFC_NCHUNKS + g];
    const uint32_t n_chunks = G.ctr[AFC_NCHUNKS + grp];
    if (G.ctr[AFC_WMODE + grp]) return;          // few problems: dp_wave_kernel's (a lane's pair of problems takes the same time whether 1 or 2000 chunks are in flight)
    while (true) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&G.ctr[AFC_CURSOR + grp], 1u);
        c = (uint32_t)__shfl((int)c, 0);
        if (c >= n_chunks) break;
        const af_chunk_t ch = G.chunks[chunk0 + c];
        const bool nodir = ch.dir_off == ~0ull;           // the direction bits did not fit: the chunk's problems are flagged, their reads take align_kernel
        bool has[2]; uint32_t tid[2] = {0, 0};
        moni_dp_task_t task[2];
        int maxq = 0, maxt = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            has[h] = (uint32_t)lane + 64u * h < ch.n;
            task[h].qlen = 0; task[h].tlen = 0; task[h].q_off = 0; task[h].t_off = 0; task[h].reserved = 0; task[h].flag = 0;
            if (has[h]) { tid[h] = G.bin_q[(size_t)ch.bin * G.bin_cap + ch.start + lane + 64 * h]; task[h] = G.tasks[tid[h]]; }
            maxq = task[h].qlen > maxq ? task[h].qlen : maxq; maxt = task[h].tlen > maxt ? task[h].tlen : maxt;
        }
        for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(maxq, o); maxq = w > maxq ? w : maxq; const int w2 = __shfl_xor(maxt, o); maxt = w2 > maxt ? w2 : maxt; }
        bool wild[2] = {nodir, nodir};          // the problem is not computed (its direction bits have no room; a wildcard base met by the plain instance - the queues keep those apart, so: never): the read takes align_kernel
        {   // query codes -> LDS, eight bases per load; bit 2 (bit 6: the high problem) marks a wildcard
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qlen = task[h].qlen, mode = task[h].reserved;
                af_bytes_t QS = af_bytes(D.reads, task[h].q_off, D.reads_limit, (mode & DP_Q_REV) != 0);
                for (int g = 0; 8 * g < maxq; ++g) {
                    const uint64_t v = 8 * g < qlen ? af_group(QS, g) : 0ull;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        uint32_t cq = dp_nt4((uint32_t)(v >> (8 * u)) & 0xFFu);
                        if ((mode & DP_Q_COMP) && cq < 4) cq = 3 - cq;
                        const bool in = 8 * g + u < qlen;
                        const uint32_t code = in ? ((cq & 3u) | (cq > 3 ? 4u : 0u)) : 0u;
                        if (!WILDC) wild[h] |= in && cq > 3;
                        if (8 * g + u < maxq) { if (h == 0) qs[8 * g + u][lane] = (uint8_t)code; else qs[8 * g + u][lane] |= (uint8_t)(code << 4); }
                    }
                }
            }
        }
        af_res_t R[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) { R[h].mqe = AF_NEG_INF; R[h].mqe_t = -1; R[h].score = AF_NEG_INF; R[h].flags = 0; }
        for (int pass = 0; pass < NP && pass * TB < maxt; ++pass) {
            const int i0 = pass * TB;
            uint32_t tp[2][NW], tn[WILDC ? NW : 1];      // target codes of the block -> registers (2 bits each); tn: the rows that hold a wildcard (tns' layout)
#pragma unroll
            for (int w = 0; w < (WILDC ? NW : 1); ++w) tn[w] = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int w = 0; w < NW; ++w) tp[h][w] = 0;
                const int tlen = task[h].tlen, mode = task[h].reserved;
                af_bytes_t TS = af_bytes(D.text, (mode & DP_T_REV) ? task[h].t_off - (uint64_t)i0 : task[h].t_off + (uint64_t)i0, D.text_limit, (mode & DP_T_REV) != 0);
#pragma unroll
                for (int g = 0; g < (TB + 7) / 8; ++g) {
                    const uint64_t v = i0 + 8 * g < tlen ? af_group(TS, g) : 0ull;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = 8 * g + u;
                        if (i < TB) {
                            const uint32_t ct = dp_nt4((uint32_t)(v >> (8 * u)) & 0xFFu);
                            const bool in = i0 + i < tlen;
                            tp[h][i >> 4] |= (in ? (ct & 3u) : 0u) << (2 * (i & 15));
                            if (WILDC) tn[i >> 4] |= (in && ct > 3 ? 1u : 0u) << (16 * h + (i & 15));
                            else wild[h] |= in && ct > 3;
                        }
                    }
                }
            }
            // A wildcard base scores sc_N against anything (ksw2: -e).  Before, such a problem sent its read to align_kernel, and 1 % of the reads
            // with an N halved the whole path's rate (profiles/r04j/bench_nrate.json)
            if (WILDC) {
#pragma unroll
                for (int w = 0; w < NW; ++w) tns[w][lane] = tn[w];
            }
            uint32_t Hc[TB], Fc[TB];
#pragma unroll
            for (int i = 0; i < TB; ++i) { Hc[i] = af_pk2(-(qo + (i0 + i + 1) * e)); Fc[i] = af_pk2(AF_NEG16); }
            uint32_t* __restrict__ dir = reinterpret_cast<uint32_t*>(G.dirs + (nodir ? 0ull : ch.dir_off) + (size_t)pass * ch.qhi * TB * 64) + lane;
            uint32_t prev_hb = af_pk2(-(qo + i0 * e));                           // H(i0 - 1, -1)
#ifdef AF_PROFILE
            if (G.dbg & 2) maxq = 0;
#endif
            const int qe0 = task[0].qlen - 1, qe1 = task[1].qlen - 1;
            for (int j = 0; j < maxq; ++j) {
                const uint32_t qb = qs[j][lane];
                // the query code of both problems spread over the 2-bit fields of a word: XOR with the target codes, fold each field to
                // one "bases differ" bit: per row a signed 1-bit extract then gives the mismatch mask of a problem
                const uint32_t qr0 = (qb & 3u) * 0x55555555u, qr1 = ((qb >> 4) & 3u) * 0x55555555u;
                uint32_t x0[NW], x1[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) { const uint32_t a0 = tp[0][w] ^ qr0, a1 = tp[1][w] ^ qr1; x0[w] = (a0 | (a0 >> 1)) & 0x55555555u; x1[w] = (a1 | (a1 >> 1)) & 0x55555555u; }
                uint32_t diag, h_up, e_run;
                if (NP > 1 && pass > 0) {
                    const uint64_t be = bnd[(size_t)j * 64];
                    h_up = (uint32_t)be; e_run = (uint32_t)(be >> 32);      // H(i0 - 1, j), E(i0 - 1, j)
                    diag = prev_hb;                                            // H(i0 - 1, j - 1)
                    prev_hb = h_up;
                } else {
                    diag = af_pk2(j == 0 ? 0 : -(qo + j * e));              // H(-1, j-1)
                    h_up = af_pk2(-(qo + (j + 1) * e));                     // H(-1, j)
                    e_run = af_pk2(AF_NEG16);
                }
                uint32_t pack = 0;
                uint32_t* __restrict__ drow = dir + (size_t)j * (TB / 4) * 64;
                const uint32_t qn2 = WILDC ? (((qb & 4u) ? 0xFFFFu : 0u) | ((qb & 0x40u) ? 0xFFFF0000u : 0u)) : 0u;      // the column's wildcard flags spread over the halves
                uint32_t tnw = 0;
#pragma unroll
                for (int i = 0; i < TB; ++i) {
                    const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)x0[i >> 4], 2 * (i & 15), 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)x1[i >> 4], 2 * (i & 15), 1);
                    const uint32_t mk = __builtin_amdgcn_perm(k1, k0, 0x05040100u);       // all ones in a half whose bases differ
                    uint32_t sc = (mk & scX2) | (~mk & scM2);
                    if (WILDC) {      // the row's wildcard bit of each problem to the sign of its half, spread over the half; or the column's
                        if ((i & 15) == 0) tnw = tns[i >> 4][lane];
                        const uint32_t nm = af_pk(af_as_s2(af_pku(af_as_u2(tnw) << (af_u2)(uint16_t)(15 - (i & 15)))) >> (af_s2)15) | qn2;
                        sc = (nm & scN2) | (~nm & sc);
                    }
                    const uint32_t h_old = Hc[i];
                    const uint32_t E = af_pk_sub(af_pk_max(af_pk_sub(h_up, qo2), e_run), e2);
                    const uint32_t F = af_pk_sub(af_pk_max(af_pk_sub(h_old, qo2), Fc[i]), e2);
                    const uint32_t zd = af_pk_add(diag, sc);
                    const uint32_t z1 = af_pk_max(zd, E), z = af_pk_max(z1, F), zq = af_pk_sub(z, qo2);
                    // sign bits: E < zd (the diagonal wins), F < z1 (it stays), E < zq, F < zq (no continuation)
                    uint32_t nb = af_pk_neg(af_pk_sub(E, zd));
                    nb |= af_pk_neg(af_pk_sub(F, z1)) << 1;
                    nb |= af_pk_neg(af_pk_sub(E, zq)) << 2;
                    nb |= af_pk_neg(af_pk_sub(F, zq)) << 3;
                    Hc[i] = z; Fc[i] = F; diag = h_old; h_up = z; e_run = E;
                    pack = (pack << 4) | nb;
                    if ((i & 3) == 3) { drow[(i >> 2) * 64] = pack; pack = 0; }
                }
                if (NP > 1) bnd[(size_t)j * 64] = (uint64_t)h_up | ((uint64_t)e_run << 32);      // (H, E) of the block's last row
                if (j == qe0 || j == qe1) {      // the last query column of one of the two problems: its mqe / score come from this column
#pragma unroll
                    for (int h = 0; h < 2; ++h) if (j == (h ? qe1 : qe0) && has[h] && !wild[h]) {
                        const int tlen = task[h].tlen;
#pragma unroll
                        for (int i = 0; i < TB; ++i) {
                            const int hv = h ? af_hi16(Hc[i]) : af_lo16(Hc[i]);
                            if (i0 + i < tlen) { if (hv > R[h].mqe) { R[h].mqe = hv; R[h].mqe_t = i0 + i; } if (i0 + i == tlen - 1) R[h].score = hv; }
                        }
                    }
                }
            }
        }
        unsigned long long cells = 0, rq = 0, cut = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) if (has[h]) {
            if (wild[h]) { R[h].mqe = AF_NEG_INF; R[h].mqe_t = -1; R[h].score = AF_NEG_INF; R[h].flags = 1; }
            else {
                const unsigned long long t_ref = (task[h].flag >> 16) ? (unsigned long long)(task[h].flag >> 16) : (unsigned long long)task[h].tlen;      // (an extension's target before the cut)
                cells += (unsigned long long)task[h].qlen * t_ref; rq += (t_ref << 32) | (unsigned long long)task[h].qlen;
                cut += (unsigned long long)task[h].qlen * (unsigned long long)task[h].tlen;
            }
            G.res[tid[h]] = R[h];
        }
        for (int o = 32; o > 0; o >>= 1) { cells += __shfl_xor(cells, o); rq += __shfl_xor(rq, o); cut += __shfl_xor(cut, o); }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CELLS]), cells);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_RBYTES]), rq >> 32);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_QBYTES]), rq & 0xFFFFFFFFull);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CUTCELLS]), cut);
            int np_run = 0;
            for (int pass = 0; pass < NP && pass * TB < maxt; ++pass) ++np_run;
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_SLOTS]), 128ull * (unsigned long long)maxq * (unsigned long long)(np_run * TB));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// dp_wave_kernel: the same recurrence, direction bits and results as dp_lane_kernel, one WAVEFRONT per pair of problems.  Lane l owns R target rows
// (R l .. R l + R - 1, counted through the passes) and walks the query one column behind lane l - 1: at step s it computes column s - l of its rows, with
// H and E of the row above (lane l - 1's last row, the column it finished in the step before) handed down by a DPP wave shift.  q + t / R steps of R cells
// where a lane of dp_lane_kernel makes q x t: a 160 x 312 global problem takes ~0.1 ms instead of ~1.9 ms.  That latency is all that counts for a queue
// with a few dozen problems (the global problems whose band is wider than AF_BANDW: ~50 per 250 000 reads; the problems with a wildcard base when few
// reads hold one) - dp_lane_kernel's launch took its 1.9 ms for a single chunk (profiles/r04m).  Per problem it issues ~4x the instructions, so
// af_chunk_kernel gives it a group only while the group has at most G.wave_max chunks.  Always keeps the wildcard planes (R rows per lane: registers are no
// concern here).  Chunk, pairing (problems w and 64 + w of a chunk in the two 16-bit halves) and direction layout are dp_lane_kernel's: traceback_kernel
// does not know which kernel wrote them.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t af_wave_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); }      // lane l gets lane l - 1's value (lane 0 keeps its own)
template <int TB, int NP, int R>
__global__ void __launch_bounds__(64) dp_wave_kernel(const af_args_t G, const uint32_t grp_a, const uint32_t grp_b) {
    static_assert(TB % R == 0 && R % 4 == 0 && R <= 16 && (TB / R) * NP <= 64, "a wavefront's lanes hold all rows of the passes");
    __shared__ uint8_t qs[AF_QCAP];          // query codes of the pair: low problem in bits 0-1 (bit 2: a wildcard), high one in bits 4-5 (bit 6)
    const int lane = threadIdx.x;
    const dp_launch_t& D = G.A.D;
    const uint32_t qo2 = af_pk2(D.qo), e2 = af_pk2(D.e), scM2 = af_pk2(D.sc_mch), scX2 = af_pk2(D.sc_mis), scN2 = af_pk2(D.sc_N);
    const int32_t qo = D.qo, e = D.e;
    for (int gi = 0; gi < 2; ++gi) {
        const uint32_t grp = gi ? grp_b : grp_a;
        if (gi && grp_b == grp_a) break;
        if (!G.ctr[AFC_WMODE + grp]) continue;
        uint32_t chunk0 = 0;
        for (uint32_t g = 0; g < grp; ++g) chunk0 += G.ctr[AFC_NCHUNKS + g];
        const uint32_t n_items = G.ctr[AFC_NCHUNKS + grp] * 64u;
        while (true) {
            uint32_t it = 0;
            if (lane == 0) it = atomicAdd(&G.ctr[AFC_CURSOR + grp], 1u);
            it = (uint32_t)__shfl((int)it, 0);
            if (it >= n_items) break;
            const af_chunk_t ch = G.chunks[chunk0 + (it >> 6)];
            const uint32_t w = it & 63u;
            if (w >= ch.n) continue;
            const bool nodir = ch.dir_off == ~0ull;
            bool has[2]; uint32_t tid[2] = {0, 0};
            moni_dp_task_t task[2];
            int maxq = 0, maxt = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                has[h] = w + 64u * h < ch.n;
                task[h].qlen = 0; task[h].tlen = 0; task[h].q_off = 0; task[h].t_off = 0; task[h].reserved = 0; task[h].flag = 0;
                if (has[h]) { tid[h] = G.bin_q[(size_t)ch.bin * G.bin_cap + ch.start + w + 64 * h]; task[h] = G.tasks[tid[h]]; }
                maxq = task[h].qlen > maxq ? task[h].qlen : maxq; maxt = task[h].tlen > maxt ? task[h].tlen : maxt;
            }
            af_res_t Rs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) { Rs[h].mqe = AF_NEG_INF; Rs[h].mqe_t = -1; Rs[h].score = AF_NEG_INF; Rs[h].flags = 0; }
            const bool fits = maxq <= AF_QCAP && maxt <= TB * NP && (uint32_t)maxq <= ch.qhi;          // (always: the queue's bounds)
            if (nodir || !fits) {
                if (lane == 0) for (int h = 0; h < 2; ++h) if (has[h]) { Rs[h].flags = 1; G.res[tid[h]] = Rs[h]; }
                continue;
            }
            // the pair's query codes -> LDS
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qlen = task[h].qlen, mode = task[h].reserved;
                for (int k = lane; k < maxq; k += 64) {
                    uint32_t code = 0;
                    if (k < qlen) {
                        uint32_t cq = dp_nt4((uint32_t)D.reads[(mode & DP_Q_REV) ? task[h].q_off - (uint64_t)k : task[h].q_off + (uint64_t)k]);
                        if ((mode & DP_Q_COMP) && cq < 4) cq = 3 - cq;
                        code = (cq & 3u) | (cq > 3 ? 4u : 0u);
                    }
                    if (h == 0) qs[k] = (uint8_t)code; else qs[k] |= (uint8_t)(code << 4);
                }
            }
            // the lane's R target rows: 2-bit codes, and the rows that hold a wildcard (low problem: bit u, high one: bit 16 + u)
            uint32_t tp[2] = {0, 0}, tnw = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tlen = task[h].tlen, mode = task[h].reserved;
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const int g = R * lane + u;
                    if (g < tlen) {
                        const uint32_t ct = dp_nt4((uint32_t)D.text[(mode & DP_T_REV) ? task[h].t_off - (uint64_t)g : task[h].t_off + (uint64_t)g]);
                        tp[h] |= (ct & 3u) << (2 * u);
                        if (ct > 3) tnw |= 1u << (16 * h + u);
                    }
                }
            }
            __syncthreads();
            uint32_t Hc[R], Fc[R];
#pragma unroll
            for (int u = 0; u < R; ++u) { Hc[u] = af_pk2(-(qo + (R * lane + u + 1) * e)); Fc[u] = af_pk2(AF_NEG16); }
            uint32_t prev_hb = af_pk2(-(qo + R * lane * e));                           // H(R lane - 1, -1)
            // the lane's rows in the direction layout: pass (R lane) / TB, row (R lane) % TB of it
            const int my_pass = (R * lane) / TB, my_i = (R * lane) % TB;
            uint32_t* __restrict__ dir = reinterpret_cast<uint32_t*>(G.dirs + ch.dir_off + (size_t)my_pass * ch.qhi * TB * 64) + (size_t)(my_i >> 2) * 64 + w;
            const int nl = (maxt + R - 1) / R, n_steps = maxq + nl - 1;
            const int qe0 = task[0].qlen - 1, qe1 = task[1].qlen - 1;
            uint32_t out_h = 0, out_e = 0;
            int bm[2] = {AF_NEG_INF, AF_NEG_INF}, bt[2] = {-1, -1}, bs[2] = {AF_NEG_INF, AF_NEG_INF};
            for (int s = 0; s < n_steps; ++s) {
                const uint32_t in_h = af_wave_shr1(out_h), in_e = af_wave_shr1(out_e);
                const int j = s - lane;
                if (j >= 0 && j < maxq && lane < nl) {
                    const uint32_t qb = qs[j];
                    const uint32_t a0 = tp[0] ^ ((qb & 3u) * 0x55555555u), a1 = tp[1] ^ (((qb >> 4) & 3u) * 0x55555555u);
                    const uint32_t x0 = (a0 | (a0 >> 1)) & 0x55555555u, x1 = (a1 | (a1 >> 1)) & 0x55555555u;
                    const uint32_t qn2 = ((qb & 4u) ? 0xFFFFu : 0u) | ((qb & 0x40u) ? 0xFFFF0000u : 0u);
                    uint32_t diag, h_up, e_run;
                    if (lane > 0) { h_up = in_h; e_run = in_e; diag = prev_hb; prev_hb = in_h; }
                    else { diag = af_pk2(j == 0 ? 0 : -(qo + j * e)); h_up = af_pk2(-(qo + (j + 1) * e)); e_run = af_pk2(AF_NEG16); }
                    uint32_t pack = 0;
                    uint32_t* __restrict__ drow = dir + (size_t)j * (TB / 4) * 64;
#pragma unroll
                    for (int u = 0; u < R; ++u) {
                        const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)x0, 2 * u, 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)x1, 2 * u, 1);
                        const uint32_t mk = __builtin_amdgcn_perm(k1, k0, 0x05040100u);
                        uint32_t sc = (mk & scX2) | (~mk & scM2);
                        const uint32_t nm = af_pk(af_as_s2(af_pku(af_as_u2(tnw) << (af_u2)(uint16_t)(15 - u))) >> (af_s2)15) | qn2;
                        sc = (nm & scN2) | (~nm & sc);
                        const uint32_t h_old = Hc[u];
                        const uint32_t E = af_pk_sub(af_pk_max(af_pk_sub(h_up, qo2), e_run), e2);
                        const uint32_t F = af_pk_sub(af_pk_max(af_pk_sub(h_old, qo2), Fc[u]), e2);
                        const uint32_t zd = af_pk_add(diag, sc);
                        const uint32_t z1 = af_pk_max(zd, E), z = af_pk_max(z1, F), zq = af_pk_sub(z, qo2);
                        uint32_t nb = af_pk_neg(af_pk_sub(E, zd));
                        nb |= af_pk_neg(af_pk_sub(F, z1)) << 1;
                        nb |= af_pk_neg(af_pk_sub(E, zq)) << 2;
                        nb |= af_pk_neg(af_pk_sub(F, zq)) << 3;
                        Hc[u] = z; Fc[u] = F; diag = h_old; h_up = z; e_run = E;
                        pack = (pack << 4) | nb;
                        if ((u & 3) == 3) { drow[(u >> 2) * 64] = pack; pack = 0; }
                    }
                    out_h = h_up; out_e = e_run;
                    if (j == qe0 || j == qe1) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) if (j == (h ? qe1 : qe0) && has[h]) {
                            const int tlen = task[h].tlen;
#pragma unroll
                            for (int u = 0; u < R; ++u) {
                                const int g = R * lane + u, hv = h ? af_hi16(Hc[u]) : af_lo16(Hc[u]);
                                if (g < tlen) { if (hv > bm[h]) { bm[h] = hv; bt[h] = g; } if (g == tlen - 1) bs[h] = hv; }
                            }
                        }
                    }
                }
            }
            // the lanes' shares of the last query column: the largest H, at the lowest row among equals; the corner
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                for (int o = 32; o > 0; o >>= 1) {
                    const int om = __shfl_xor(bm[h], o), ot = __shfl_xor(bt[h], o), os = __shfl_xor(bs[h], o);
                    if (om > bm[h] || (om == bm[h] && ot >= 0 && (bt[h] < 0 || ot < bt[h]))) { bm[h] = om; bt[h] = ot; }
                    bs[h] = os > bs[h] ? os : bs[h];
                }
                Rs[h].mqe = bm[h]; Rs[h].mqe_t = bt[h]; Rs[h].score = bs[h];
            }
            if (lane == 0) {
                unsigned long long cells = 0, rb = 0, qb_ = 0, cut = 0;
                for (int h = 0; h < 2; ++h) if (has[h]) {
                    const unsigned long long t_ref = (task[h].flag >> 16) ? (unsigned long long)(task[h].flag >> 16) : (unsigned long long)task[h].tlen;
                    cells += (unsigned long long)task[h].qlen * t_ref; rb += t_ref; qb_ += (unsigned long long)task[h].qlen;
                    cut += (unsigned long long)task[h].qlen * (unsigned long long)task[h].tlen;
                    G.res[tid[h]] = Rs[h];
                }
                atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CELLS]), cells);
                atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_RBYTES]), rb);
                atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_QBYTES]), qb_);
                atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CUTCELLS]), cut);
                atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_SLOTS]), 2ull * (unsigned long long)maxq * (unsigned long long)(nl * R));
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// dp_band_kernel: a GLOBAL problem (ksw_extz2_sse without EXTZ_ONLY: score = H(tlen - 1, qlen - 1), traceback from that corner; call site
// aligner_ksw2.hpp:3015) over the W diagonals global_band_kernel has shown to hold every cell of every best path.  Lane = two problems (16-bit halves, as
// dp_lane_kernel).  Slot k of a lane's registers holds, at query column j, the cell of row i = j + dlo + k: its diagonal neighbour is the slot's own value of
// the column before, its left neighbour slot k + 1 of the column before, its upper neighbour slot k - 1 of this column - so a column is one pass over the
// slots in place, with nothing to shift.  What lies outside the band reads as "minus infinity"; the matrix's own border (row -1, column -1) needs no special
// case: H(-1, -1) = 0 is set, and the recurrence then produces -(q + k e) along row -1 by itself (an insertion run), column -1 is written at the start.
// Cells of best paths get exactly the values and direction bits of the full matrix (a neighbour that a best path could come from lies in the band; one that
// lies outside loses strictly).  160 x 16 cell slots per problem where the full-matrix kernel steps through 160 x 208.
// ------------------------------------------------------------------------------------------------------------------------------
#define AF_BTCAP (AF_QCAP + 2 * AF_BANDW)
template <int W>
__global__ void __launch_bounds__(64) dp_band_kernel(const af_args_t G, const uint32_t grp) {
    __shared__ uint8_t qs[AF_QCAP / 2][64];      // query codes of the lane's two problems, two positions per byte: position x in nibble x & 1 of byte x >> 1, the low problem's
    __shared__ uint8_t ts[AF_BTCAP / 2][64];     // code in the nibble's bits 0-1, the high one's in bits 2-3; target codes likewise (17 KB: two blocks per SIMD - a byte per position was 35 KB, one)
    static_assert(AF_QCAP % 2 == 0 && AF_BTCAP % 2 == 0, "two positions per byte");
    static_assert(W % 4 == 0 && W <= 16, "a lane's target window is one 32-bit word of 2-bit codes");
    const int lane = threadIdx.x;
    const dp_launch_t& D = G.A.D;
    const uint32_t qo2 = af_pk2(D.qo), e2 = af_pk2(D.e), scM2 = af_pk2(D.sc_mch), scX2 = af_pk2(D.sc_mis), neg2 = af_pk2(AF_NEG16);
    const int32_t qo = D.qo, e = D.e;
    uint32_t chunk0 = 0;
    for (uint32_t g = 0; g < grp; ++g) chunk0 += G.ctr[AFC_NCHUNKS + g];
    const uint32_t n_chunks = G.ctr[AFC_NCHUNKS + grp];
    while (true) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&G.ctr[AFC_CURSOR + grp], 1u);
        c = (uint32_t)__shfl((int)c, 0);
        if (c >= n_chunks) break;
        const af_chunk_t ch = G.chunks[chunk0 + c];
        const bool nodir = ch.dir_off == ~0ull;
        bool has[2]; uint32_t tid[2] = {0, 0};
        moni_dp_task_t task[2];
        int dlo[2] = {0, 0};
        int maxq = 0, maxt = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            has[h] = (uint32_t)lane + 64u * h < ch.n;
            task[h].qlen = 0; task[h].tlen = 0; task[h].q_off = 0; task[h].t_off = 0; task[h].reserved = 0; task[h].flag = 0;
            if (has[h]) { tid[h] = G.bin_q[(size_t)ch.bin * G.bin_cap + ch.start + lane + 64 * h]; task[h] = G.tasks[tid[h]]; dlo[h] = (int)(int16_t)((uint32_t)task[h].reserved >> 16); }
            maxq = task[h].qlen > maxq ? task[h].qlen : maxq; maxt = task[h].tlen > maxt ? task[h].tlen : maxt;
        }
        for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(maxq, o); maxq = w > maxq ? w : maxq; const int w2 = __shfl_xor(maxt, o); maxt = w2 > maxt ? w2 : maxt; }
        if (maxq > AF_QCAP) maxq = AF_QCAP;
        if (maxt > AF_BTCAP) maxt = AF_BTCAP;
        bool wild[2] = {nodir, nodir};
#pragma unroll
        for (int h = 0; h < 2; ++h) {          // both sequences of both problems -> LDS, eight bases per load
            if (task[h].qlen > AF_QCAP || task[h].tlen > AF_BTCAP) wild[h] = true;          // (cannot happen: the band is narrower than the two lengths differ)
            const int qlen = task[h].qlen, tlen = task[h].tlen, mode = task[h].reserved;
            af_bytes_t QS = af_bytes(D.reads, task[h].q_off, D.reads_limit, (mode & DP_Q_REV) != 0);
            for (int g = 0; 8 * g < maxq; ++g) {
                const uint64_t v = 8 * g < qlen ? af_group(QS, g) : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    uint32_t cq = dp_nt4((uint32_t)(v >> (8 * u)) & 0xFFu);
                    if ((mode & DP_Q_COMP) && cq < 4) cq = 3 - cq;
                    const bool in = 8 * g + u < qlen;
                    wild[h] |= in && cq > 3;
                    if (8 * g + u < maxq) {
                        const uint32_t c = in ? (cq & 3u) : 0u;
                        if (h == 0 && (u & 1) == 0) qs[4 * g + (u >> 1)][lane] = (uint8_t)c; else qs[4 * g + (u >> 1)][lane] |= (uint8_t)(c << (2 * h + 4 * (u & 1)));
                    }
                }
            }
            af_bytes_t TS = af_bytes(D.text, task[h].t_off, D.text_limit, (mode & DP_T_REV) != 0);
            for (int g = 0; 8 * g < maxt; ++g) {
                const uint64_t v = 8 * g < tlen ? af_group(TS, g) : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t ct = dp_nt4((uint32_t)(v >> (8 * u)) & 0xFFu);
                    const bool in = 8 * g + u < tlen;
                    wild[h] |= in && ct > 3;
                    if (8 * g + u < maxt) {
                        const uint32_t c = in ? (ct & 3u) : 0u;
                        if (h == 0 && (u & 1) == 0) ts[4 * g + (u >> 1)][lane] = (uint8_t)c; else ts[4 * g + (u >> 1)][lane] |= (uint8_t)(c << (2 * h + 4 * (u & 1)));
                    }
                }
            }
        }
        // target code of row i of problem h (rows outside the staged ones: any code - they hold no cell that counts)
        auto tcode = [&](int i, int h) -> uint32_t { const int r = i < 0 ? 0 : i >= maxt ? maxt - 1 : i; return maxt > 0 ? ((uint32_t)ts[r >> 1][lane] >> (4 * (r & 1) + 2 * h)) & 3u : 0u; };
        // column -1: slot k holds row -1 + dlo + k
        uint32_t Hb[W], Fb[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int r0 = -1 + dlo[0] + k, r1 = -1 + dlo[1] + k;
            const int v0 = r0 == -1 ? 0 : r0 >= 0 ? -(qo + (r0 + 1) * e) : AF_NEG16, v1 = r1 == -1 ? 0 : r1 >= 0 ? -(qo + (r1 + 1) * e) : AF_NEG16;
            Hb[k] = ((uint32_t)v0 & 0xFFFFu) | ((uint32_t)v1 << 16);
            Fb[k] = neg2;
        }
        uint32_t tw[2] = {0, 0};          // the target codes of the slots' rows at the current column, 2 bits per slot
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < W; ++k) tw[h] |= tcode(dlo[h] + k, h) << (2 * k);
        uint32_t* __restrict__ dir = reinterpret_cast<uint32_t*>(G.dirs + (nodir ? 0ull : ch.dir_off)) + lane;
        const int qe0 = task[0].qlen - 1, qe1 = task[1].qlen - 1;
        const int ks0 = (task[0].tlen - 1) - qe0 - dlo[0], ks1 = (task[1].tlen - 1) - qe1 - dlo[1];          // the corner's slot at the last column
        const bool ext0 = (task[0].flag & DP_EZ_EXTZ_ONLY) != 0, ext1 = (task[1].flag & DP_EZ_EXTZ_ONLY) != 0;      // an extension: the last column's maximum and its first row
        int score[2] = {AF_NEG_INF, AF_NEG_INF}, mqe[2] = {AF_NEG_INF, AF_NEG_INF}, mqe_t[2] = {-1, -1};
        for (int j = 0; j < maxq; ++j) {
            const uint32_t qb = ((uint32_t)qs[j >> 1][lane] >> (4 * (j & 1))) & 0xFu;
            const uint32_t a0 = tw[0] ^ ((qb & 3u) * 0x55555555u), a1 = tw[1] ^ (((qb >> 2) & 3u) * 0x55555555u);
            const uint32_t x0 = (a0 | (a0 >> 1)) & 0x55555555u, x1 = (a1 | (a1 >> 1)) & 0x55555555u;
            uint32_t h_up = neg2, e_run = neg2, pack = 0;
            uint32_t* __restrict__ drow = dir + (size_t)j * (W / 4) * 64;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)x0, 2 * k, 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)x1, 2 * k, 1);
                const uint32_t mk = __builtin_amdgcn_perm(k1, k0, 0x05040100u);
                const uint32_t sc = (mk & scX2) | (~mk & scM2);
                const uint32_t h_old = Hb[k];
                const uint32_t left_h = k + 1 < W ? Hb[k + 1] : neg2, left_f = k + 1 < W ? Fb[k + 1] : neg2;
                const uint32_t E = af_pk_sub(af_pk_max(af_pk_sub(h_up, qo2), e_run), e2);
                const uint32_t F = af_pk_sub(af_pk_max(af_pk_sub(left_h, qo2), left_f), e2);
                const uint32_t zd = af_pk_add(h_old, sc);
                const uint32_t z1 = af_pk_max(zd, E), z = af_pk_max(z1, F), zq = af_pk_sub(z, qo2);
                uint32_t nb = af_pk_neg(af_pk_sub(E, zd));
                nb |= af_pk_neg(af_pk_sub(F, z1)) << 1;
                nb |= af_pk_neg(af_pk_sub(E, zq)) << 2;
                nb |= af_pk_neg(af_pk_sub(F, zq)) << 3;
                Hb[k] = z; Fb[k] = F; h_up = z; e_run = E;
                pack = (pack << 4) | nb;
                if ((k & 3) == 3) { drow[(k >> 2) * 64] = pack; pack = 0; }
            }
            if (j == qe0 || j == qe1) {
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    if (j == qe0) {
                        const int hv = af_lo16(Hb[k]), i = qe0 + dlo[0] + k;
                        if (k == ks0) score[0] = hv;
                        if (ext0 && i >= 0 && i < task[0].tlen && hv > mqe[0]) { mqe[0] = hv; mqe_t[0] = i; }
                    }
                    if (j == qe1) {
                        const int hv = af_hi16(Hb[k]), i = qe1 + dlo[1] + k;
                        if (k == ks1) score[1] = hv;
                        if (ext1 && i >= 0 && i < task[1].tlen && hv > mqe[1]) { mqe[1] = hv; mqe_t[1] = i; }
                    }
                }
            }
            // the rows move down by one with the next column
            tw[0] = (tw[0] >> 2) | (tcode(j + 1 + dlo[0] + W - 1, 0) << (2 * (W - 1)));
            tw[1] = (tw[1] >> 2) | (tcode(j + 1 + dlo[1] + W - 1, 1) << (2 * (W - 1)));
        }
        unsigned long long cells = 0, rq = 0, cut = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) if (has[h]) {
            const bool ext = (h ? ext1 : ext0);
            af_res_t R; R.mqe = ext ? mqe[h] : AF_NEG_INF; R.mqe_t = ext ? mqe_t[h] : -1; R.score = ext ? AF_NEG_INF : score[h]; R.flags = 0;
            if (wild[h] || (ext ? mqe_t[h] < 0 : score[h] == AF_NEG_INF)) { R.mqe = AF_NEG_INF; R.mqe_t = -1; R.score = AF_NEG_INF; R.flags = 1; }
            else {
                const unsigned long long t_ref = (task[h].flag >> 16) ? (unsigned long long)(task[h].flag >> 16) : (unsigned long long)task[h].tlen;      // (an extension's target before the cut)
                cells += (unsigned long long)task[h].qlen * t_ref; rq += (t_ref << 32) | (unsigned long long)task[h].qlen;
                cut += (unsigned long long)task[h].qlen * (unsigned long long)(task[h].tlen < W ? task[h].tlen : W);
            }
            G.res[tid[h]] = R;
        }
        for (int o = 32; o > 0; o >>= 1) { cells += __shfl_xor(cells, o); rq += __shfl_xor(rq, o); cut += __shfl_xor(cut, o); }
        if (lane == 0) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CELLS]), cells);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_RBYTES]), rq >> 32);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_QBYTES]), rq & 0xFFFFFFFFull);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_CUTCELLS]), cut);
            atomicAdd(reinterpret_cast<unsigned long long*>(&G.ctr[AFC_SLOTS]), 128ull * (unsigned long long)maxq * (unsigned long long)W);
        }
        __syncthreads();
    }
}

// where the direction byte of cell (i, j) of a task is: its chunk, its lane there, the target block of i
struct af_dirs_t { const uint8_t* base; uint32_t tb, half; uint64_t pass_stride; };      // half: which 16-bit half of the words holds the task
__device__ __forceinline__ af_dirs_t af_dir_of(const af_args_t& G, uint32_t bin, uint32_t pos_in_bin) {
    const uint32_t grp = af_grp_of_bin(bin);
    const uint32_t b1 = af_grp_b1(grp);
    uint32_t ci = 0;
    for (uint32_t g = 0; g < grp; ++g) ci += G.ctr[AFC_NCHUNKS + g];
    for (uint32_t b2 = bin + 1; b2 < b1; ++b2) ci += ((G.ctr[AFC_BINS + b2] < G.bin_cap ? G.ctr[AFC_BINS + b2] : G.bin_cap) + 127) >> 7;        // chunks are laid out from the group's last bin down
    ci += pos_in_bin >> 7;
    const af_chunk_t ch = G.chunks[ci];
    af_dirs_t X;
    X.tb = af_grp_tb(grp);
    X.pass_stride = (uint64_t)ch.qhi * X.tb * 64;
    X.base = G.dirs + ch.dir_off + (size_t)(pos_in_bin & 63) * 4;
    X.half = (pos_in_bin >> 6) & 1u;
    return X;
}

// the window of a chain from its extensions (aligner_ksw2.hpp:2852-2886)
__device__ __forceinline__ void af_window(const af_cand_t& C, const af_anchor_t* an, uint32_t m, int lc_t, int rc_t, uint64_t& ref_pos, uint64_t& ref_len) {      // an: the chain's anchors
    const af_anchor_t first = an[0], last = an[C.n_an - 1];
    const uint64_t mem_pos = first.occ, mem_len = last.occ + last.len - mem_pos;
    const uint64_t lq = (uint64_t)(int64_t)(first.idx > 0 ? lc_t + 1 : 0);
    const uint64_t rcs_len = m - ((uint64_t)last.idx + last.len);
    const uint64_t rq = (uint64_t)(int64_t)(rcs_len > 0 ? rc_t + 1 : 0);
    ref_pos = lq > mem_pos ? 0 : mem_pos - lq; ref_len = lq + mem_len + rq;
}

// The diagonals d = i - j a best-scoring path of a GLOBAL problem (corner to corner, aligner_ksw2.hpp:3009-3015) can touch.  A lower bound of the
// score from the alignment "diagonal 0, one gap of |tlen - qlen| where it pays most, diagonal tlen - qlen" (one pass over the two sequences);
// an upper bound for any path that touches diagonal D outside [min(0, delta), max(0, delta)]: it holds at least 2 D - delta (above) or
// delta - 2 D (below) gap bases in at least two gaps, and at most min(qlen, tlen) diagonal steps.  Diagonals whose upper bound lies strictly below
// the lower bound hold no cell of any best path - nor of any path that ties with one.
__device__ __forceinline__ void af_global_band(const dp_launch_t& D, const moni_dp_task_t& T, int& dmin, int& dmax, bool* saw_wild) {
    const int q = T.qlen, t = T.tlen, delta = t - q, ad = delta < 0 ? -delta : delta, n = q < t ? q : t;
    const bool qrev = (T.reserved & DP_Q_REV) != 0, qcomp = (T.reserved & DP_Q_COMP) != 0, trev = (T.reserved & DP_T_REV) != 0;
    const uint64_t qsh = delta < 0 ? (uint64_t)ad : 0ull, tsh = delta > 0 ? (uint64_t)ad : 0ull;
    af_bytes_t Q0 = af_bytes(D.reads, T.q_off, D.reads_limit, qrev), Q1 = af_bytes(D.reads, qrev ? T.q_off - qsh : T.q_off + qsh, D.reads_limit, qrev);
    af_bytes_t T0 = af_bytes(D.text, T.t_off, D.text_limit, trev), T1 = af_bytes(D.text, trev ? T.t_off - tsh : T.t_off + tsh, D.text_limit, trev);
    int tot = 0, A = 0, maxA = 0;
    bool wildc = false;
    for (int g = 0; 8 * g < n; ++g) {
        const uint64_t vq0 = af_group(Q0, g), vt0 = af_group(T0, g);
        const uint64_t vq1 = delta < 0 ? af_group(Q1, g) : vq0, vt1 = delta > 0 ? af_group(T1, g) : vt0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (8 * g + u >= n) break;
            uint32_t a = dp_nt4((uint32_t)(vq0 >> (8 * u)) & 0xFFu), b = dp_nt4((uint32_t)(vq1 >> (8 * u)) & 0xFFu);
            if (qcomp) { if (a < 4) a = 3 - a; if (b < 4) b = 3 - b; }
            const uint32_t c = dp_nt4((uint32_t)(vt0 >> (8 * u)) & 0xFFu), d = dp_nt4((uint32_t)(vt1 >> (8 * u)) & 0xFFu);
            const int s0 = (a > 3 || c > 3) ? D.sc_N : a == c ? D.sc_mch : D.sc_mis;          // diagonal 0
            const int s1 = (b > 3 || d > 3) ? D.sc_N : b == d ? D.sc_mch : D.sc_mis;          // diagonal delta
            wildc = wildc || a > 3 || b > 3 || c > 3 || d > 3;
            tot += s1; A += s0 - s1; maxA = A > maxA ? A : maxA;
        }
    }
    if (saw_wild) *saw_wild = wildc;          // (only the positions the two diagonals pair: a wildcard elsewhere in the longer sequence is found by the kernel that stages it)
    const int lb = tot + maxA - (delta ? D.qo + D.e * ad : 0);
    const int G = D.e > 0 ? (D.sc_mch * n - 2 * D.qo - lb) : 0x3FFFFFFF;          // e x (gap bases a path can afford)
    dmin = delta < 0 ? delta : 0; dmax = delta > 0 ? delta : 0;
    if (G >= 0 && D.e > 0) {
        const int gb = G / D.e;
        const int up = (gb + delta) >> 1, dn = -((gb - delta) >> 1);
        dmax = up > dmax ? up : dmax; dmin = dn < dmin ? dn : dmin;
    }
    if (D.e <= 0) { dmin = -q; dmax = t; }
}

// The same for an EXTENSION (EXTZ_ONLY: mqe = max_i H(i, qlen - 1), the first row that holds it, traceback from there): the path ends anywhere in the last
// column, so only the lower bound's end is fixed - the diagonal itself, H(qlen - 1, qlen - 1) >= the sum of its match / mismatch scores.  A path that touches
// diagonal D > 0 deletes at least D target bases (<= qlen sc_mch - qo - D e), one that touches D < 0 inserts at least |D| query bases and has that many fewer
// diagonal steps (<= (qlen - |D|) sc_mch - qo - |D| e).  Rows that hold the maximum are reached by such paths: they lie in the band too.
__device__ __forceinline__ void af_ext_band(const dp_launch_t& D, const moni_dp_task_t& T, int& dmin, int& dmax) {
    const int q = T.qlen, t = T.tlen;
    dmin = -q; dmax = t;
    if (t < q || D.e <= 0) return;                         // (the target ends before the diagonal does: no bound from it)
    const bool qrev = (T.reserved & DP_Q_REV) != 0, qcomp = (T.reserved & DP_Q_COMP) != 0, trev = (T.reserved & DP_T_REV) != 0;
    af_bytes_t Q0 = af_bytes(D.reads, T.q_off, D.reads_limit, qrev), T0 = af_bytes(D.text, T.t_off, D.text_limit, trev);
    int lb = 0;
    for (int g = 0; 8 * g < q; ++g) {
        const uint64_t vq = af_group(Q0, g), vt = af_group(T0, g);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (8 * g + u >= q) break;
            uint32_t a = dp_nt4((uint32_t)(vq >> (8 * u)) & 0xFFu);
            if (qcomp && a < 4) a = 3 - a;
            const uint32_t c = dp_nt4((uint32_t)(vt >> (8 * u)) & 0xFFu);
            lb += (a > 3 || c > 3) ? D.sc_N : a == c ? D.sc_mch : D.sc_mis;
        }
    }
    const int G = D.sc_mch * q - D.qo - lb;
    dmin = 0; dmax = 0;
    if (G >= 0) { dmax = G / D.e; dmin = -(G / (D.e + D.sc_mch)); }
}

// The mismatches on the main diagonal of a tile problem from the 2-bit forms the seeding stage keeps (32 bases per word: the diagonal of a 60-base extension is
// two words of the pattern against two of the text, where the byte form walks 8 bases per load through up to four streams - band_tasks_kernel was bound by
// those dependent loads: 2.5 ms per 1 M reads).  The query of a problem is a stretch [a, a + qlen) of the strand-resolved pattern of its read (the task's
// byte offset and mode give read, strand and a); a left extension runs backwards through pattern and text alike, so its diagonal pairs are those of the two
// stretches aligned at their ENDS.  n: positions compared (from the stretches' common start - or end, for a left extension).  False: a byte outside
// A / C / G / T on either side, or a stretch that is not where the formulas put it (the caller then takes the byte form).
__device__ __forceinline__ bool af_diag_mm2(const af_args_t& G, const moni_dp_task_t& T, uint64_t read, uint64_t off, uint32_t m, int n, uint32_t& mm) {
    const dp_launch_t& D = G.A.D;
    const int q = T.qlen;
    if (!G.pat || !G.text2 || n <= 0 || n > q || n > T.tlen) return false;
    const bool comp = (T.reserved & DP_Q_COMP) != 0, rev = (T.reserved & DP_Q_REV) != 0;
    const bool reversed = comp ? !rev : rev;
    const uint64_t d = T.q_off - off;
    uint64_t a = !comp ? (rev ? d - (uint64_t)q + 1 : d) : (rev ? (uint64_t)m - 1 - d : (uint64_t)m - (uint64_t)q - d);
    if (a + (uint64_t)q > m) return false;
    if (reversed) a += (uint64_t)(q - n);                                       // the last n positions of the stretch
    if (reversed && T.t_off + 1 < (uint64_t)n) return false;
    const uint64_t t0 = reversed ? T.t_off - (uint64_t)n + 1 : T.t_off;          // first text position compared
    if (t0 + (uint64_t)n > D.n_text) return false;
    { const uint64_t b0 = t0 >> G.exc_sh, b1 = (t0 + (uint64_t)n - 1) >> G.exc_sh; if (((G.exc[b0 >> 5] >> (b0 & 31u)) | (G.exc[b1 >> 5] >> (b1 & 31u))) & 1u) return false; }
    const uint64_t task = 2 * read + (comp ? 1u : 0u);
    const uint64_t pb = ws_pat_base(G.blk, task), lb = ws_block_len(G.blk, task);
    const uint64_t n2w = (lb + 31) / 32, cb = pb + 64u * ((lb + 7) / 8), mb = cb + 64u * n2w;
    uint64_t bad = 0;
    mm = 0;
    for (int k = 0; k < n; k += 32) {
        const uint64_t pa = a + (uint64_t)k, ta = t0 + (uint64_t)k;
        const uint32_t ps = 2u * (uint32_t)(pa & 31u), ts = 2u * (uint32_t)(ta & 31u);
        // (the word behind a stretch's last one may lie behind the read's words - the next region of the workspace: such a load made every stretch that ends
        // in the read's last word - a right extension, mostly - look marked, and a third of the extensions went to the full tile without a bound: profiles/r05f)
        const bool more = ps && (pa >> 5) + 1 < n2w;
        const uint64_t p0 = G.pat[cb + (pa >> 5) * 64u], p1 = more ? G.pat[cb + ((pa >> 5) + 1) * 64u] : 0ull;
        const uint64_t w0 = G.text2[ta >> 5], w1 = ts ? G.text2[(ta >> 5) + 1] : 0ull;
        bad |= G.pat[mb + (pa >> 5) * 64u] | (more ? G.pat[mb + ((pa >> 5) + 1) * 64u] : 0ull);          // (whole words: a mark beside the stretch costs only the byte form)
        const uint64_t pw = ps ? (p0 >> ps) | (p1 << (64u - ps)) : p0, tw = ts ? (w0 >> ts) | (w1 << (64u - ts)) : w0;
        uint64_t x = pw ^ tw;
        x = (x | (x >> 1)) & 0x5555555555555555ull;
        const int left = n - k;
        if (left < 32) x &= (1ull << (2 * left)) - 1ull;
        mm += (uint32_t)__popcll(x);
    }
    return bad == 0;
}
// af_global_band's bound for a gap fill whose lengths differ, from the 2-bit forms: the alignment "diagonal 0, one gap of |tlen - qlen| where it pays most, diagonal
// tlen - qlen" scores tot + maxA - (qo + e |delta|), tot = the score of diagonal delta over its n = min(qlen, tlen) pairs, maxA = the largest prefix sum of
// (diagonal 0's score - diagonal delta's) over the pairs in the problem's order.  In read / text coordinates the two diagonals are the stretches aligned at their
// STARTS and at their ENDS (which is which, and the order of the pairs, turn around for a problem that runs backwards); per 32 pairs one XOR each, and the prefix
// maximum walks the set bits of the two mismatch words - a handful.  (Since round 4's first banding these problems - 129 000 per 250 000 reads, 40 % of the tile
// kernel's - had gone to the full tile: the byte form of this bound was what made band_tasks_kernel slow.)  False: a stretch that is not where the formulas put it.
__device__ __forceinline__ bool af_two_piece2(const af_args_t& G, const moni_dp_task_t& T, uint64_t read, uint64_t off, uint32_t m, int& dmin, int& dmax) {
    const dp_launch_t& D = G.A.D;
    const int q = T.qlen, t = T.tlen, delta = t - q, ad = delta < 0 ? -delta : delta, n = q < t ? q : t;
    if (!G.pat || !G.text2 || n <= 0 || n > 128 || D.e <= 0) return false;
    const bool comp = (T.reserved & DP_Q_COMP) != 0, rev = (T.reserved & DP_Q_REV) != 0, trev = (T.reserved & DP_T_REV) != 0;
    const bool reversed = comp ? !rev : rev;
    if (reversed != trev) return false;                                         // (query and target of a problem run the same way)
    const uint64_t d = T.q_off - off;
    const uint64_t a = !comp ? (rev ? d - (uint64_t)q + 1 : d) : (rev ? (uint64_t)m - 1 - d : (uint64_t)m - (uint64_t)q - d);      // the query is pattern[a, a + q) of the read's strand-resolved pattern
    if (a + (uint64_t)q > m) return false;
    if (reversed && T.t_off + 1 < (uint64_t)t) return false;
    const uint64_t tlo = reversed ? T.t_off - (uint64_t)t + 1 : T.t_off;          // the target is text[tlo, tlo + t)
    if (tlo + (uint64_t)t > D.n_text) return false;
    { const uint64_t b0 = tlo >> G.exc_sh, b1 = (tlo + (uint64_t)t - 1) >> G.exc_sh; if (((G.exc[b0 >> 5] >> (b0 & 31u)) | (G.exc[b1 >> 5] >> (b1 & 31u))) & 1u) return false; }
    const uint64_t task = 2 * read + (comp ? 1u : 0u);
    if (G.pflag && G.pflag[task]) return false;                                 // (a byte outside A / C / G / T somewhere in the read: the tile)
    const uint64_t pb = ws_pat_base(G.blk, task), lb = ws_block_len(G.blk, task);
    const uint64_t cb = pb + 64u * ((lb + 7) / 8), n2w = (lb + 31) / 32, n_tw = (D.n_text + 31) >> 5;
    auto pbits = [&](uint64_t first) -> uint64_t {
        const uint64_t i = first >> 5; const uint32_t sh = 2u * (uint32_t)(first & 31u);
        const uint64_t lo = i < n2w ? G.pat[cb + i * 64u] : 0ull;
        if (!sh) return lo;
        return (lo >> sh) | ((i + 1 < n2w ? G.pat[cb + (i + 1) * 64u] : 0ull) << (64u - sh));
    };
    auto tbits = [&](uint64_t first) -> uint64_t {
        const uint64_t i = first >> 5; const uint32_t sh = 2u * (uint32_t)(first & 31u);
        const uint64_t lo = i < n_tw ? G.text2[i] : 0ull;
        if (!sh) return lo;
        return (lo >> sh) | ((i + 1 < n_tw ? G.text2[i + 1] : 0ull) << (64u - sh));
    };
    // mismatch words of the two alignments, 32 pairs per word (bit 2 u of word c: pair 32 c + u), in read / text coordinates
    uint64_t xs[4] = {0, 0, 0, 0}, xe[4] = {0, 0, 0, 0};
    const uint64_t pe0 = a + (uint64_t)(q - n), te0 = tlo + (uint64_t)(t - n);
#pragma unroll
    for (int c = 0; c < 4; ++c) if (32 * c < n) {
        const int left = n - 32 * c;
        const uint64_t lowm = left < 32 ? (1ull << (2 * left)) - 1ull : ~0ull;
        uint64_t x = pbits(a + 32u * c) ^ tbits(tlo + 32u * c);
        xs[c] = (x | (x >> 1)) & 0x5555555555555555ull & lowm;
        x = pbits(pe0 + 32u * c) ^ tbits(te0 + 32u * c);
        xe[c] = (x | (x >> 1)) & 0x5555555555555555ull & lowm;
    }
    // diagonal 0 pairs the problem's k-th elements: the starts for a problem that runs forwards, the ends for one that runs backwards; diagonal delta the other
    int pc1 = 0, run = 0, best = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) pc1 += (int)__popcll(reversed ? xs[c] : xe[c]);
    if (!reversed) {
        for (int c = 0; c < 4 && 32 * c < n; ++c) {
            uint64_t m0 = xs[c], m1 = xe[c], any = m0 | m1;
            while (any) { const uint64_t bit = any & (0ull - any); run += ((m1 & bit) ? 1 : 0) - ((m0 & bit) ? 1 : 0); best = run > best ? run : best; any ^= bit; }
        }
    } else {
        for (int c = 3; c >= 0; --c) if (32 * c < n) {
            uint64_t m0 = xe[c], m1 = xs[c], any = m0 | m1;
            while (any) { const uint64_t bit = 1ull << (63 - __builtin_clzll(any)); run += ((m1 & bit) ? 1 : 0) - ((m0 & bit) ? 1 : 0); best = run > best ? run : best; any ^= bit; }
        }
    }
    const int dm = D.sc_mch - D.sc_mis;
    const int tot = D.sc_mch * n - dm * pc1, maxA = dm * best;
    const int lbnd = tot + maxA - (D.qo + D.e * ad);
    const int Gq = D.sc_mch * n - 2 * D.qo - lbnd;
    dmin = delta < 0 ? delta : 0; dmax = delta > 0 ? delta : 0;
    if (Gq >= 0) {
        const int gb = Gq / D.e;
        const int up = (gb + delta) >> 1, dn = -((gb - delta) >> 1);
        dmax = up > dmax ? up : dmax; dmin = dn < dmin ? dn : dmin;
    }
    return true;
}
// af_ext_band / af_global_band from that count.  An extension: the diagonal's score bounds mqe from below.  A gap fill or global problem whose two lengths
// agree: the diagonal is itself a corner-to-corner path; lengths that differ: af_two_piece2.
__device__ __forceinline__ bool af_tile_band2(const af_args_t& G, const moni_dp_task_t& T, uint64_t read, uint64_t off, uint32_t m, int& dmin, int& dmax) {
    const dp_launch_t& D = G.A.D;
    const int q = T.qlen, t = T.tlen;
    if (D.e <= 0) return false;
    uint32_t mm = 0;
    if (T.flag & DP_EZ_EXTZ_ONLY) {
        dmin = -q; dmax = t;
        if (t < q) {
            // More query than target (the reference cuts an extension's target at ext_len rows whatever the read's part is): every end cell (i, q - 1) lies on a diagonal
            // <= t - q = -ad.  Lower bound of mqe: the diagonal over the t target rows, then the ad remaining query bases as one insertion.  A path whose lowest
            // diagonal is -(ad + y) inserts at least ad + y bases, so it has at most t - y diagonal steps: t mch - y (mch + e) - (qo + ad e) at best; one that rises to
            // diagonal d > 0 deletes d bases and inserts d + ad: (t - d) mch - 2 qo - (2 d + ad) e at best.  Diagonals whose best lies below the bound hold no cell of a
            // best path.  (22 000 extensions per 250 000 reads, the tile kernel's longest: with the bound nearly all of them fit dp_band_kernel's 16 diagonals.  A
            // 32-diagonal instance of that kernel for the rest was built and measured: its launch cost more than the 2 700 problems it took saved - profiles/r05i.)
            if (!af_diag_mm2(G, T, read, off, m, t, mm)) return false;
            const int ad = q - t, dm = (D.sc_mch - D.sc_mis) * (int)mm;
            dmin = -ad - dm / (D.e + D.sc_mch);
            dmax = dm - D.qo > 0 ? (dm - D.qo) / (D.sc_mch + 2 * D.e) : 0;
            return true;
        }
        if (!af_diag_mm2(G, T, read, off, m, q, mm)) return false;
        const int G_ = (D.sc_mch - D.sc_mis) * (int)mm - D.qo;          // qlen sc_mch - qo - (the diagonal's score)
        dmin = 0; dmax = 0;
        if (G_ >= 0) { dmax = G_ / D.e; dmin = -(G_ / (D.e + D.sc_mch)); }
        return true;
    }
    if (q != t) return af_two_piece2(G, T, read, off, m, dmin, dmax);
    if (!af_diag_mm2(G, T, read, off, m, q, mm)) return false;
    const int G_ = (D.sc_mch - D.sc_mis) * (int)mm - 2 * D.qo;          // min(qlen, tlen) sc_mch - 2 qo - (the diagonal's score): af_global_band with delta = 0
    dmin = 0; dmax = 0;
    if (G_ >= 0) { const int gb = G_ / D.e; dmax = gb >> 1; dmin = -(gb >> 1); }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------------
// global_task_kernel: chains with overlapping anchors are scored by one global alignment of the whole read against the window
// their extensions give (aligner_ksw2.hpp:2984-2996, 3009-3015); one lane per read queues those problems
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) global_task_kernel(const af_args_t G) {
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const ak_args_t& A = G.A;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    // lanes stay converged (the counters are bumped once per wave, not once per lane: one address takes only ~50 M atomics/s)
    bool active = r_in < A.n_reads;
    af_plan_t* PLp = active ? &G.plans[r_in] : nullptr;
    if (active && PLp->status != AF_ST_CAND) active = false;
    const uint64_t r = A.read_lo + r_in;          // the plan's read, or (paired-end) its pair
    uint64_t off = 0; uint32_t m = 0, n_cand = 0;
    if (active) { n_cand = PLp->n_cand; if (!G.pe) { off = A.offs[r]; m = (uint32_t)(A.offs[r + 1] - off); } }
    uint32_t why = AF_WHY_N;
    uint32_t max_cand = n_cand;
    for (int d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)max_cand, d); max_cand = o > max_cand ? o : max_cand; }
    for (uint32_t c = 0; c < max_cand; ++c) {
        bool need = false;
        uint64_t ref_pos = 0, ref_len = 0;
        af_cand_t* Cp = nullptr;
        if (active && why == AF_WHY_N && c < n_cand) {
            Cp = &PLp->cand[c];
            if (G.pe) { const uint64_t rr = 2 * r + (Cp->pad & 1u); off = A.offs[rr]; m = (uint32_t)(A.offs[rr + 1] - off); }      // the mate this share of the chain belongs to
            if (Cp->overlap) {
                uint32_t t = Cp->task0;
                int lc_t = -1, rc_t = -1;
                if (Cp->has_lc) { const af_res_t R = G.res[t++]; if (R.flags) why = AF_WHY_WILDCARD; lc_t = R.mqe_t; }
                if (Cp->has_rc) { const af_res_t R = G.res[t++]; if (R.flags) why = AF_WHY_WILDCARD; rc_t = R.mqe_t; }
                af_window(*Cp, PLp->an + Cp->an0, m, lc_t, rc_t, ref_pos, ref_len);
                if (why == AF_WHY_N) {
                    Cp->gtask = ~0u;
                    if (ac_valid(A.P, ref_pos, ref_len)) {                    // else: scored INT32_MIN whatever the DP says; never the final chain
                        if (ref_len == 0 || ref_len > (uint64_t)AF_GPASS * AF_GBLK) why = AF_WHY_TASK_SIZE;
                        else need = true;
                    }
                }
            }
        }
        const unsigned long long nm = __ballot(need);
        if (nm == 0) continue;
        uint32_t base = 0;
        if (lane == __ffsll((long long)nm) - 1) base = atomicAdd(&G.ctr[AFC_TASKS], (uint32_t)__popcll(nm));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)nm) - 1);
        const uint32_t tid = base + (uint32_t)__popcll(nm & lt_mask);
        if (need && tid >= G.task_cap) { why = AF_WHY_CAPACITY; need = false; }
        if (need) {          // the problem's record; global_band_kernel (one lane per problem) bounds its diagonals and queues it
            moni_dp_task_t T;
            if (!Cp->strand) { T.q_off = off; T.reserved = DP_Q_READS | DP_T_TEXT; } else { T.q_off = off + m - 1; T.reserved = DP_Q_READS | DP_T_TEXT | DP_Q_REV | DP_Q_COMP; }
            T.t_off = ref_pos; T.qlen = (int32_t)m; T.tlen = (int32_t)ref_len; T.flag = DP_EZ_RIGHT;
            if (!G.pflag || G.pflag[2 * (G.pe ? 2 * r + (Cp->pad & 1u) : r) + (Cp->strand ? 1u : 0u)]) T.reserved |= AF_T_WILDQ;
            G.tasks[tid] = T;
            af_res_t R0; R0.mqe = AF_NEG_INF; R0.mqe_t = -1; R0.score = AF_NEG_INF; R0.flags = 1;          // (stays so if a queue turns out to be full: the read then takes align_kernel)
            G.res[tid] = R0;
            Cp->gtask = tid;
        }
    }
    if (active && why != AF_WHY_N) { PLp->status = AF_ST_FALLBACK; G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)r; atomicAdd(&G.ctr[AFC_WHY + why], 1u); }
}

// global_band_kernel: one lane per global problem (they lie behind the reads' task slots: [AFC_GT0, AFC_TASKS)).  The band of diagonals its optimal paths can
// touch (af_global_band: a few for a read that differs from the window by substitutions - 77 % of the benchmark's global problems need at most 4 diagonals,
// 99.8 % at most 16; profiles/r04e) decides the kernel: dp_band_kernel keeps AF_BANDW diagonals in registers and steps through qlen x AF_BANDW cells where
// the full matrix has qlen x tlen; a problem with a wider band takes the full-matrix kernel as before.  The band's first diagonal rides in the upper half of the task's `reserved` word.
__global__ void __launch_bounds__(256) global_band_kernel(const af_args_t G) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t t0 = (uint32_t)G.A.n_reads * AF_MAX_TASKS_READ, t1 = G.ctr[AFC_TASKS] < G.task_cap ? G.ctr[AFC_TASKS] : G.task_cap;
    const uint32_t tid = t0 + blockIdx.x * 256 + threadIdx.x;
    bool need = tid < t1;
    uint32_t bin = 0, hb = 0;
    moni_dp_task_t T;
    if (need) {
        T = G.tasks[tid];
        int dlo, dhi;
        bool saw_wild = false;
        af_global_band(G.A.D, T, dlo, dhi, &saw_wild);
        const int W = dhi - dlo + 1;
        hb = (uint32_t)(W / 4 < 13 ? W / 4 : 13);
        const bool wq = saw_wild || (T.reserved & AF_T_WILDQ) || af_target_exc(G, T);          // a wildcard base may lie in the problem: the full-matrix kernel's WILDC instance scores it
        const bool band = W <= AF_BANDW && !wq && !(G.dbg & 0x10000u);          // (MONI_AF_DBG=65536: every global problem through the full-matrix kernel)
        if (band) {          // the band widened to AF_BANDW diagonals around what is needed (never beyond what the matrix has)
            const int spare = AF_BANDW - W;
            dlo -= spare / 2;
            T.reserved = (T.reserved & 0xFFFF) | (int)(((uint32_t)dlo & 0xFFFFu) << 16);
            G.tasks[tid].reserved = T.reserved;
        }
        bin = band ? AF_BIN_GBAND + af_large_bin(T.qlen) : (wq ? AF_BIN_GWILD : AF_BIN_GLOBAL) + (uint32_t)((T.qlen - 1) >> 4);
    }
    uint32_t at = 0;
    unsigned long long rest = __ballot(need);
    while (rest) {                                     // one bump per distinct bin of the wave
        const int lead = __ffsll((long long)rest) - 1;
        const uint32_t b = (uint32_t)__shfl((int)bin, lead);
        const unsigned long long same = __ballot(need && bin == b);
        uint32_t a0 = 0;
        if (lane == lead) a0 = atomicAdd(&G.ctr[AFC_BINS + b], (uint32_t)__popcll(same));
        a0 = (uint32_t)__shfl((int)a0, lead);
        if (need && bin == b) at = a0 + (uint32_t)__popcll(same & lt_mask);
        rest &= ~same;
    }
    if (need && at < G.bin_cap) { G.bin_q[(size_t)bin * G.bin_cap + at] = tid; G.task_pos[tid] = at | (bin << AF_POS_BITS); }
    rest = __ballot(need);                             // the histogram of band widths, one bump per distinct class of the wave (one atomic per problem on the
    while (rest) {                                     // counter of the narrowest class - three in four fall into it - took 2.4 ms per 1 M reads: profiles/r04g)
        const int lead = __ffsll((long long)rest) - 1;
        const uint32_t b = (uint32_t)__shfl((int)hb, lead);
        const unsigned long long same = __ballot(need && hb == b);
        if (lane == lead) atomicAdd(&G.ctr[AFC_BANDH + b], (uint32_t)__popcll(same));
        rest &= ~same;
    }
}

// band_tasks_kernel: one lane per problem of the provisional list (bin_tasks_kernel: extensions and gap fills of the large tile).  Its band of diagonals
// (af_ext_band / af_global_band) - AF_BANDW or fewer: dp_band_kernel's queue of its query length, the band's first diagonal in the task record; else the
// tile's.  (Bounding the band inside bin_tasks_kernel, whose lanes are task SLOTS - one in eight in use, a block's rounds one after the other - took
// 7.3 ms per 1 M reads: profiles/r04h.)
__global__ void __launch_bounds__(256) band_tasks_kernel(const af_args_t G) {
    __shared__ uint32_t cnt[AF_NBIN], base[AF_NBIN];
    const uint32_t tid = threadIdx.x;
    const uint32_t n = G.ctr[AFC_BINS + AF_BIN_PROV] < G.bin_cap ? G.ctr[AFC_BINS + AF_BIN_PROV] : G.bin_cap;
    if (blockIdx.x * 256u >= n) return;                    // (block-uniform)
    if (tid < AF_NBIN) cnt[tid] = 0;
    __syncthreads();
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    const bool need = x < n;
    uint32_t bin = 0, id = 0, local = 0;
    if (need) {
        id = G.bin_q[(size_t)AF_BIN_PROV * G.bin_cap + x];
        const moni_dp_task_t t = G.tasks[id];
        int dlo, dhi;
        {
            // the read the problem belongs to: its plan (id / slots per read), paired-end: the mate whose bytes hold the query
            uint64_t rd = G.A.read_lo + id / AF_MAX_TASKS_READ;
            if (G.pe) { rd *= 2; if (t.q_off >= G.A.offs[rd + 1]) ++rd; }
            const uint64_t off = G.A.offs[rd];
            // (no bound from the 2-bit forms - a gap fill whose lengths differ, a byte outside A / C / G / T: the tile takes the problem.  One lane in the byte form's
            // dependent loads would hold up its whole wavefront)
            const bool found = af_tile_band2(G, t, rd, off, (uint32_t)(G.A.offs[rd + 1] - off), dlo, dhi);
            if (!found) { dlo = -t.qlen; dhi = t.tlen; }
#if defined(AF_CUTS)
            {   // (experiments: the tile problems by kind and by what the bound says)
                const bool ext = (t.flag & DP_EZ_EXTZ_ONLY) != 0;
                const uint32_t kind = ext ? 0u : t.qlen == t.tlen ? 1u : 2u;
                const uint32_t what = !found ? 3u : (ext && t.tlen < t.qlen) ? 2u : (dhi - dlo + 1 <= AF_BANDW) ? 0u : 1u;
                atomicAdd(&G.ctr[180 + 4 * kind + what], 1u);
            }
#endif
        }
        const int W = dhi - dlo + 1;
        bin = af_large_bin(t.qlen);
#if defined(AF_CUTS)

#endif
        if (W <= AF_BANDW) {
            dlo -= (AF_BANDW - W) / 2;
            G.tasks[id].reserved = (t.reserved & 0xFFFF) | (int)(((uint32_t)dlo & 0xFFFFu) << 16);
            bin += AF_BIN_BAND;
        }
        local = atomicAdd(&cnt[bin], 1u);
    }
    __syncthreads();
    // one bump of a queue's counter per block and queue, all of a block's bumps side by side (a wavefront that bumped the ~30 queues its problems fall into one
    // after the other, each time waiting for the count to come back, made this kernel take 2.3 ms per 1 M reads: profiles/r04k)
    if (tid < AF_NBIN && cnt[tid]) base[tid] = atomicAdd(&G.ctr[AFC_BINS + tid], cnt[tid]);
    __syncthreads();
    if (need) {
        const uint32_t at = base[bin] + local;
        if (at < G.bin_cap) { G.bin_q[(size_t)bin * G.bin_cap + at] = id; G.task_pos[id] = at | (bin << AF_POS_BITS); }
        else { af_res_t R0; R0.mqe = AF_NEG_INF; R0.mqe_t = -1; R0.score = AF_NEG_INF; R0.flags = 1; G.res[id] = R0; }      // the queue is full: a result "not computed" sends the read to align_kernel (select_kernel)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// select_kernel: the results come back into the selection loop (aligner_ksw2.hpp:436-474, 528-548, 2852-2886, 2975-2999)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) select_kernel(const af_args_t G) {
    const uint64_t r_in = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const ak_args_t& A = G.A;
    if (r_in >= A.n_reads) return;
    af_plan_t& PL = G.plans[r_in];
    if (PL.status != AF_ST_CAND) return;
    const ac_params_t& P = A.P;
    const uint64_t r = A.read_lo + r_in;
    const uint32_t m = (uint32_t)(A.offs[r + 1] - A.offs[r]);
    const uint32_t n_cand = PL.n_cand;
    int32_t max_score = 0; uint32_t n_alt = 0;
    struct best_t { int32_t score; uint64_t lft; uint64_t idx; } best[AF_MAX_CAND + 2];
    uint32_t n_best = 0;
    bool fallback = false;
    uint32_t why = AF_WHY_WILDCARD;
    for (uint32_t c = 0; c < n_cand && !fallback; ++c) {
        af_cand_t& C = PL.cand[c];
        uint32_t t = C.task0;
        int score_lc = 0, score_rc = 0, lc_t = -1, rc_t = -1;
        if (C.has_lc) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; score_lc = R.mqe; lc_t = R.mqe_t; }
        if (C.has_rc) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; score_rc = R.mqe; rc_t = R.mqe_t; }
        const uint32_t n_an = C.n_an;
        const af_anchor_t* const AN = PL.an + C.an0;
        uint64_t ref_pos, ref_len;
        af_window(C, AN, m, lc_t, rc_t, ref_pos, ref_len);
        int32_t score;
        if (C.overlap) {
            score = INT32_MIN;                                              // gtask == ~0u: the window is not inside one sequence
            if (C.gtask != ~0u) { const af_res_t R = G.res[C.gtask]; fallback |= R.flags != 0; score = R.score; }
        } else {
            uint32_t sc = (uint32_t)score_lc + (uint32_t)score_rc;
            for (uint32_t k = 1; k < n_an; ++k) {
                const af_anchor_t ap = AN[k - 1];
                int32_t gs = ap.gap_val;                                        // AF_GAP_NONE: 0
                if (ap.gap_kind == AF_GAP_INS) gs = af_ins_score(P, (uint64_t)(uint16_t)ap.gap_val);
                else if (ap.gap_kind == AF_GAP_TASK) { const af_res_t R = G.res[t++]; fallback |= R.flags != 0; gs = R.score; }
                sc += (uint32_t)((uint64_t)ap.len * (uint64_t)(int64_t)P.smatch + (uint64_t)(int64_t)gs);
            }
            sc += (uint32_t)((uint64_t)AN[n_an - 1].len * (uint64_t)(int64_t)P.smatch);
            score = (int32_t)sc;
            if (!ac_valid(P, ref_pos, ref_len)) score = INT32_MIN;
        }
        C.score = score;
        // check_max_score + the best_scores update (ac_absorb), W.i = C.chain_idx
        const uint64_t lft = ac_lift(P, ref_pos);
        if (score > max_score) { max_score = score; n_alt = 0; }
        else if (score == max_score) { PL.alt_pos[n_alt] = ref_pos; PL.alt_score[n_alt] = score; ++n_alt; }      // n_alt <= n_cand
        uint64_t wi = C.chain_idx;
        bool replaced = false;
        for (uint32_t j = 0; j < n_best; ++j) {
            const uint64_t bl = best[j].lft;
            const uint64_t d = bl > lft ? bl - lft : lft - bl;
            if (d < P.region_dist) {
                if (score > best[j].score) {
                    if (replaced) { best[j].score = 0; best[j].lft = 0; best[j].idx = wi - 1; }
                    else { best[j].score = score; best[j].lft = lft; best[j].idx = wi; wi++; replaced = true; }
                } else { j = n_best; replaced = true; wi++; }
            }
        }
        if (!replaced) { best[n_best].score = score; best[n_best].lft = lft; best[n_best].idx = wi; ++n_best; wi++; }
        if (wi != (uint64_t)C.chain_idx + 1 && !fallback) { fallback = true; why = AF_WHY_LOOP; }        // the loop skipped a chain because of this score: the run-ahead list is not the loop's
    }
    uint32_t status = AF_ST_UNALIGNED;
    uint32_t final_c = 0;
    int32_t score2 = 0;
    if (!fallback) {
        while (n_best < 2) { best[n_best].score = 0; best[n_best].lft = 0; best[n_best].idx = PL.n_chains; ++n_best; }
        // std::sort(greater<tuple>) of at most AF_MAX_CAND + 2 elements: the insertion-sort range of std::sort (n <= 16), i.e. a stable sort
        for (uint32_t i = 1; i < n_best; ++i) {
            const best_t v = best[i];
            uint32_t k = i;
            auto gt = [](const best_t& x, const best_t& y) { return x.score > y.score || (x.score == y.score && (x.lft > y.lft || (x.lft == y.lft && x.idx > y.idx))); };
            while (k > 0 && gt(v, best[k - 1])) { best[k] = best[k - 1]; --k; }
            best[k] = v;
        }
        if (best[0].score >= PL.min_score) {
            score2 = best[1].score;
            const uint64_t fc = best[0].idx;
            int found = -1;
            for (uint32_t c = 0; c < n_cand; ++c) if (PL.cand[c].chain_idx == fc) found = (int)c;
            if (fc < PL.n_chains && found >= 0 && PL.cand[found].score >= PL.min_score) { status = AF_ST_FINAL; final_c = (uint32_t)found; }
            else if (fc < PL.n_chains && found < 0) { fallback = true; why = AF_WHY_LOOP; }      // cannot happen (a positive score belongs to a scored chain); stay exact
        }
    }
    if (!fallback && status == AF_ST_FINAL) {
        // the final call of chain_score repeats the chain's problems with EXTZ_ONLY | RIGHT and a traceback (aligner_ksw2.hpp:2062-2076);
        // the direction bytes are there already; an extension must reach the query end for its traceback to start at (mqe_t, qlen - 1)
        af_cand_t& C = PL.cand[final_c];
        uint32_t t = C.task0;
        int lc_t = -1, rc_t = -1;
        for (uint32_t k = 0; k < (uint32_t)C.has_lc + C.has_rc; ++k) {
            const af_res_t R = G.res[t + k];
            const moni_dp_task_t T = G.tasks[t + k];
            const int32_t bound = (T.qlen < T.tlen ? T.qlen : T.tlen) * (int32_t)A.D.sc_mch;             // ez->max <= sc_mch * min(qlen, tlen)
            if (!C.overlap && !(R.mqe + A.D.end_bonus > bound) && !fallback) { fallback = true; why = AF_WHY_REACH_END; }
            if (k == 0 && C.has_lc) lc_t = R.mqe_t; else rc_t = R.mqe_t;
        }
        af_window(C, PL.an + C.an0, m, lc_t, rc_t, PL.ref_pos, PL.ref_len);
        const uint32_t n_tb = C.overlap ? 1u : (uint32_t)C.has_lc + C.has_rc + C.n_gap_tasks;
        PL.pad = (uint16_t)(C.strand | (n_tb << 8));          // finish_wave_kernel starts its fetches from the plan's header alone
        PL.pad2 = (uint32_t)C.an0 | ((uint32_t)C.n_an << 16);
        if (!fallback && n_tb) {
            const uint32_t tb0 = atomicAdd(&G.ctr[AFC_TRACED], n_tb);
            if (tb0 + n_tb > G.tb_cap) { fallback = true; why = AF_WHY_CAPACITY; }
            else { PL.tb0 = tb0; if (C.overlap) G.tb_task[tb0] = C.gtask; else for (uint32_t k = 0; k < n_tb; ++k) G.tb_task[tb0 + k] = t + k; }
        }
    }
    if (fallback) { PL.status = AF_ST_FALLBACK; G.fb_list[atomicAdd(G.fb_n, 1u)] = (uint32_t)r; atomicAdd(&G.ctr[AFC_WHY + why], 1u); return; }
    PL.status = (uint8_t)status; PL.final_cand = (uint8_t)final_c; PL.score2 = score2; PL.n_alt = (uint8_t)n_alt;
}

// ------------------------------------------------------------------------------------------------------------------------------
// traceback_kernel: ksw_backtrack (is_rot) over the direction bytes of one problem per lane
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) traceback_kernel(const af_args_t G) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x;
    if (x >= G.ctr[AFC_TRACED] || x >= G.tb_cap) return;
    const uint32_t tid = G.tb_task[x];
    const moni_dp_task_t T = G.tasks[tid];
    const af_res_t R = G.res[tid];
    const uint32_t bin = G.task_pos[tid] >> AF_POS_BITS, pos = G.task_pos[tid] & ((1u << AF_POS_BITS) - 1);
    af_tb_t& O = G.tb[x];
    const af_dirs_t X = af_dir_of(G, bin, pos);
    const uint32_t tb = X.tb;
    int i, j;
    if (T.flag & DP_EZ_EXTZ_ONLY) { i = R.mqe_t; j = T.qlen - 1; } else { i = T.tlen - 1; j = T.qlen - 1; }
    uint32_t n = 0, cur_op = 0xFu, cur_len = 0;
    bool ovf = false;
    auto push = [&](uint32_t op, uint32_t len) {
        if (op == cur_op) { cur_len += len; return; }
        if (cur_op != 0xFu) { if (n < AF_TB_CIG) O.ops[n++] = cur_len << 4 | cur_op; else ovf = true; }
        cur_op = op; cur_len = len;
    };
    int state = 0;
    const bool banded = af_bin_banded(bin);                          // dp_band_kernel's layout: slot = i - j - first diagonal of the band (the task's flag word)
    const int band_lo = (int)(int16_t)((uint32_t)T.reserved >> 16);
    while (i >= 0 && j >= 0) {
        uint32_t ps = (uint32_t)i / tb, ii = (uint32_t)i - ps * tb;
        if (banded) { const int kk = i - j - band_lo; if (kk < 0 || kk >= (int)tb) { ovf = true; break; } ps = 0; ii = (uint32_t)kk; }      // (a best path never leaves the band)
        const uint32_t word = *reinterpret_cast<const uint32_t*>(X.base + ps * X.pass_stride + ((size_t)j * (tb / 4) + (size_t)(ii >> 2)) * 256);
        const uint32_t nb = (word >> (16 * X.half + 4 * (3 - (ii & 3)))) & 0xFu;      // sign bits of dp_lane_kernel: diagonal beats E, that beats F, E ends, F ends
        const uint32_t tmp = ((nb & 2u) ? ((nb & 1u) ? 0u : 1u) : 2u) | ((nb & 4u) ? 0u : 0x08u) | ((nb & 8u) ? 0u : 0x10u);      // ksw2's direction byte
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (state == 0) { push(0, 1); --i; --j; }
        else if (state == 1 || state == 3) { push(2, 1); --i; }
        else { push(1, 1); --j; }
    }
    if (i >= 0) push(2, (uint32_t)(i + 1));
    if (j >= 0) push(1, (uint32_t)(j + 1));
    if (cur_op != 0xFu) { if (n < AF_TB_CIG) O.ops[n++] = cur_len << 4 | cur_op; else ovf = true; }
    O.n_ops = ovf ? ~0u : n;
}

// ------------------------------------------------------------------------------------------------------------------------------
// finish_kernel: stitched CIGAR of the final chain (aligner_ksw2.hpp:3049-3108) and the record / SAM line (ak_write_record)
// ------------------------------------------------------------------------------------------------------------------------------
struct af_fin_t { uint32_t cig[AF_FIN_CIG]; uint32_t lcig[AF_FIN_LCIG]; uint64_t md_tmp[AK_MD_CAP / 8]; uint64_t txt_tmp[AK_TXT_CAP / 8]; };

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) finish_kernel(const af_args_t G) {
    const ak_args_t& A = G.A;
    af_fin_t& S = *reinterpret_cast<af_fin_t*>(G.fin_scratch + ((size_t)blockIdx.x * 64 + threadIdx.x) * G.fin_stride);
    for (uint64_t r_in = (uint64_t)blockIdx.x * 64 + threadIdx.x; r_in < A.n_reads; r_in += (uint64_t)gridDim.x * 64) {
        af_plan_t& PL = G.plans[r_in];
        if (PL.status == AF_ST_FALLBACK) continue;
        const uint64_t r = A.read_lo + r_in;
        ak_final_t V;
        V.off = A.offs[r]; V.m = (uint32_t)(A.offs[r + 1] - A.offs[r]); V.strand = 0; V.ref_pos = 0; V.score = 0; V.score2 = 0;
        V.cigar = S.cig; V.n_cigar = 0; V.alt_pos = PL.alt_pos; V.alt_score = PL.alt_score; V.n_alt = 0; V.aligned = 0; V.overflow = 0;
        if (PL.status == AF_ST_FINAL) {
            const af_cand_t& C = PL.cand[PL.final_cand];
            uint32_t n = 0; bool ovf = false;
            auto push = [&](uint32_t op) { if (n < AF_FIN_CIG) S.cig[n++] = op; else ovf = true; };
            auto push_merge_first = [&](uint32_t op, bool first) { if (first && (op & 0xf) == 0 && n > 0) S.cig[n - 1] += op; else push(op); };
            uint32_t tbx = PL.tb0;
            if (C.overlap) {             // the CIGAR of the global realignment (aligner_ksw2.hpp:3009-3020)
                const af_tb_t& T = G.tb[tbx];
                if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push(T.ops[T.n_ops - 1 - k]);
            } else {
                if (C.has_lc) {              // the CIGAR of the reversed problem, reversed again: the traceback's own order
                    const af_tb_t& T = G.tb[tbx++];
                    if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push(T.ops[k]);
                }
                const uint32_t rc_x = C.has_rc ? tbx++ : 0u;
                const af_anchor_t* const AN = PL.an + C.an0;
                for (uint32_t j = 0; j < C.n_an; ++j) {
                    const uint32_t mlen = AN[j].len;
                    if (n > 0 && (S.cig[n - 1] & 0xf) == 0) S.cig[n - 1] += mlen << 4; else push(mlen << 4);
                    if (j + 1 < C.n_an) {
                        const af_anchor_t g = AN[j];
                        if (g.gap_kind == AF_GAP_TASK) {
                            const af_tb_t& T = G.tb[tbx++];
                            if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push_merge_first(T.ops[T.n_ops - 1 - k], k == 0);
                        } else if (g.gap_kind == AF_GAP_INS) push_merge_first(((uint32_t)(uint16_t)g.gap_val << 4) | 1u, true);
                        else if (g.gap_kind == AF_GAP_DEL0) push_merge_first(2u, true);
                        else if (g.gap_kind == AF_GAP_1X1) push_merge_first(1u << 4, true);
                    }
                }
                if (C.has_rc) {
                    const af_tb_t& T = G.tb[rc_x];
                    if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) push_merge_first(T.ops[T.n_ops - 1 - k], k == 0);
                }
            }
            if (ovf) {       // the host pipeline redoes the read (align_kernel runs beside this kernel, its list is closed)
                atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CIGAR], 1u);
                moni_aln_rec_t rec;
                rec.status = 2u; rec.strand = C.strand; rec.ref_pos = 0; rec.score = 0; rec.score2 = 0;
                rec.n_cigar = 0; rec.n_alt = 0; rec.cigar_off = 0; rec.alt_off = 0; rec.nm = 0; rec.md_len = 0; rec.md_off = 0; rec.txt_len = 0; rec.lift_nm = 0; rec.txt_off = 0;
                A.recs[r_in] = rec;
                if (A.dev_len) { A.dev_len[r_in] = 0; A.dev_off[r_in] = 0; atomicAdd(&A.dev_sum[0], 1ull); }
                continue;
            }
            V.strand = C.strand; V.ref_pos = PL.ref_pos; V.score = C.score; V.score2 = PL.score2; V.n_cigar = n; V.n_alt = PL.n_alt; V.aligned = 1;
        }
        ak_write_record(A, V, S.md_tmp, S.txt_tmp, S.lcig, AF_FIN_LCIG, r_in, r);      // a record that does not fit the pools is marked for the host pipeline, as in align_kernel
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// finish_wave_kernel: the same work as finish_kernel, one wavefront per read, the SAM line assembled in LDS and written out with
// coalesced stores.  The lanes stage the chain record, the traceback records and the strand-oriented read; lane 0 stitches the CIGAR
// and lifts it (aligner_ksw2.hpp:3049-3108, 3133-3175); MD / NM (write_MD_core, sam.hpp:249-287) come from 64-column ballots, every
// mismatching lane writing its own MD item.  The line itself (sam.hpp:144-188) is not spelled character by character: lane 0 lists
// its SEGMENTS (a literal, a decimal number, a sequence name, SEQ, QUAL, reference bases ...: ~60 per line, a few instructions each)
// and then all lanes render the bytes of the line side by side, each finding its segment by binary search.  MAPQ (mapq.hpp:146-184) in
// IEEE double operations in the reference's order.  A line, CIGAR or MD beyond the LDS staging goes to the host pipeline.
// ------------------------------------------------------------------------------------------------------------------------------
#define AFS_LINE 1280            // bytes of one line in LDS
#define AFS_MAXSEG 96            // segments of one line (a CIGAR and the MD string are one segment each)
#define AFS_MAXMD 256            // MD items: one per mismatch or deletion (with the count of matches before it) + the closing count
#define AFS_CIG 64               // operations of the stitched CIGAR ...
#define AFS_LCIG 128             // ... and of the lifted one
#define AFW_NAMES 1024           // sequence names kept in LDS when they fit (else they are read from HBM)
#define AFW_NSEQ 126
#define AFW_TB_WORDS (AF_TB_CIG + 1)
static __device__ const char afs_lit[] = "\t" "\t*\t0\t0\t" "\tAS:i:" "\tNM:i:" "\tZS:i:" "\tMD:Z:" "\tOA:Z:" ",+," ",-," "," ";" "\tAA:Z:" "\n" "\t4\t*\t0\t255\t*\t*\t0\t0\t" "*" "^" "MIDNSHP=X" "ACGTN";
static_assert(sizeof(afs_lit) == 89, "literal table");
enum { LT_TAB = 0, LT_MATE = 1, LT_AS = 8, LT_NM = 14, LT_ZS = 20, LT_MD = 26, LT_OA = 32, LT_PLUS = 38, LT_MINUS = 41, LT_COMMA = 44, LT_SEMI = 45, LT_AA = 46, LT_NL = 52,
       LT_UNAL = 53, LT_STAR = 72, LT_CARET = 73, LT_OPS = 74, LT_BASES = 83 };
enum { SK_LIT = 0, SK_NUM, SK_NEG, SK_NAME, SK_RNAME, SK_SEQ, SK_QUAL, SK_CIG, SK_MD };
struct af_finw_t {
    uint8_t line[AFS_LINE];
    uint8_t seq[AF_MAX_READ];            // the read in alignment orientation (ASCII, kpbseq.h:120-137 complement)
    uint32_t cig[AFS_CIG], lcig[AFS_LCIG];
    uint32_t n_cig, n_lcig, ovf, n_seg, total, seq_at, qual_at;      // seq_at / qual_at: where SEQ and QUAL start in the line (QUAL absent: ~0)
    uint8_t names[AFW_NAMES]; uint16_t name_off[AFW_NSEQ + 2];      // the index's sequence names (kernel lifetime)
    af_cand_t cand; af_anchor_t an[AF_FIN_AN];                       // the final chain's record, its first anchors and the alternatives, fetched by all lanes at once
    uint64_t alt_pos[AF_MAX_CAND]; int32_t alt_score[AF_MAX_CAND]; uint32_t alt_sid[AF_MAX_CAND], alt_p1[AF_MAX_CAND];
    uint32_t md_item[AFS_MAXMD];         // type (2 bits: 0 closing count, 1 mismatch, 2 deletion) | matches before it << 2 | mismatch: reference base << 12;
                                         // deletion: length << 12 | offset of its first base in the reference window << 21
    uint16_t md_off[AFS_MAXMD + 1];      // where an item's text starts in the MD string
    uint16_t cig_off[AFS_CIG + 1], lcig_off[AFS_LCIG + 1];      // the same for the operations of the two CIGAR strings
    uint16_t seg_off[AFS_MAXSEG + 1]; uint8_t seg_kind[AFS_MAXSEG]; uint32_t seg_val[AFS_MAXSEG];
};

__device__ __forceinline__ uint32_t afs_ndig(uint32_t u) {
    return 1u + (u >= 10u) + (u >= 100u) + (u >= 1000u) + (u >= 10000u) + (u >= 100000u) + (u >= 1000000u) + (u >= 10000000u) + (u >= 100000000u) + (u >= 1000000000u);
}
// lane 0: one more segment of the line
template <class LT>
__device__ __forceinline__ void afs_push(LT& L, uint32_t& n, uint32_t& p, uint32_t kind, uint32_t val, uint32_t len) {
    if (len == 0) return;
    if (n < AFS_MAXSEG) { L.seg_off[n] = (uint16_t)(p < 0xFFFFu ? p : 0xFFFFu); L.seg_kind[n] = (uint8_t)kind; L.seg_val[n] = val; }
    ++n; p += len;
}
template <class LT>
__device__ __forceinline__ void afs_lits(LT& L, uint32_t& n, uint32_t& p, uint32_t at, uint32_t len) { afs_push(L, n, p, SK_LIT, at, len); }
template <class LT>
__device__ __forceinline__ void afs_num(LT& L, uint32_t& n, uint32_t& p, int v) {
    if (v < 0) { const uint32_t u = 0u - (uint32_t)v; afs_push(L, n, p, SK_NEG, u, afs_ndig(u) + 1); }
    else afs_push(L, n, p, SK_NUM, (uint32_t)v, afs_ndig((uint32_t)v));
}
// a CIGAR string as one segment: the offsets of its operations' texts (number + letter) for the renderer
template <class LT>
__device__ __forceinline__ void afs_cigar(LT& L, uint32_t& n, uint32_t& p, const uint32_t* cg, uint32_t nc, uint16_t* offs, uint32_t which) {
    uint32_t q = 0;
    for (uint32_t k = 0; k < nc; ++k) { offs[k] = (uint16_t)q; q += afs_ndig(cg[k] >> 4) + 1u; }
    offs[nc] = (uint16_t)q;
    afs_push(L, n, p, SK_CIG, which | (nc << 1), q);
}

// MD / NM of one CIGAR over the window that starts at text position t0 (write_MD_core).  All lanes call it (uniform control flow).
// items: the MD string as a list in L.md_item (a mismatching lane writes its own item: the matches before it and its base); n_items
// counts them (AFS_MAXMD + 1: something did not fit, the caller sends the read to the host pipeline).  Returns NM (uniform).
// LT: holds the read in alignment orientation (seq) and the item list (md_item): af_finw_t, or one mate of a pair (pe_lines.hip).
template <class LT>
__device__ __forceinline__ int afs_md(const dp_launch_t& D, LT& L, const uint32_t* cg, uint32_t n_cig, uint64_t t0, bool items, uint32_t& n_items,
                                      const uint8_t* win) {          // win: the window's nt4 codes in LDS (or nullptr: read the text)
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    int NM = 0; uint32_t l_MD = 0, ni = 0;
    bool bad = false;
    uint64_t t = t0; uint32_t q = 0;
    for (uint32_t i = 0; i < n_cig; ++i) {
        const uint32_t op = cg[i] & 0xf, len = cg[i] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (uint32_t k0 = 0; k0 < len; k0 += 64) {
                const uint32_t k = k0 + lane;
                uint32_t tc = 0; bool mis = false;
                if (k < len) { const uint64_t a = t + k; tc = win ? (uint32_t)win[a - t0] : dp_nt4(a < D.n_text ? D.text[a] : 0u); mis = dp_nt4(L.seq[q + k]) != tc; }
                const unsigned long long bal = __ballot(mis);
                const uint32_t span = len - k0 < 64 ? len - k0 : 64;
                if (bal) {
                    const uint32_t cnt = (uint32_t)__popcll(bal);
                    if (items && mis) {
                        const unsigned long long prev = bal & lt_mask;
                        const uint32_t run = prev ? (uint32_t)lane - (63u - (uint32_t)__clzll((long long)prev)) - 1u : (uint32_t)lane + l_MD;
                        const uint32_t at = ni + (uint32_t)__popcll(prev);
                        if (at < AFS_MAXMD) L.md_item[at] = 1u | ((run & 0x3FFu) << 2) | (tc << 12);
                    }
                    if (l_MD + 63u > 0x3FFu) bad = true;                  // a count of matches beyond the item's 10 bits
                    ni += cnt; NM += (int)cnt;
                    l_MD = span - 1u - (63u - (uint32_t)__clzll((long long)bal));
                } else l_MD += span;
            }
            q += len; t += len;
        } else if (op == 1) { q += len; NM += (int)len; }
        else if (op == 2) {
            if (items) {
                if (lane == 0 && ni < AFS_MAXMD) L.md_item[ni] = 2u | ((l_MD & 0x3FFu) << 2) | ((len & 0x1FFu) << 12) | ((uint32_t)(t - t0) << 21);
                if (l_MD > 0x3FFu || len > 0x1FFu || (t - t0) > 0x7FFu) bad = true;
            }
            ++ni;
            l_MD = 0; t += len; NM += (int)len;
        } else if (op == 3) t += len;
    }
    if (items && l_MD > 0) { if (lane == 0 && ni < AFS_MAXMD) L.md_item[ni] = (l_MD & 0x3FFu) << 2; if (l_MD > 0x3FFu) bad = true; ++ni; }
    n_items = (bad || ni > AFS_MAXMD) ? AFS_MAXMD + 1 : ni;
    return NM;
}

// traced problem k / anchor j of the final chain: the first AF_FIN_TB / AF_FIN_AN are staged in LDS, the rest is read where it lies in HBM
__device__ __forceinline__ const af_tb_t* afw_tb(const uint32_t* tbs, const af_tb_t* tb, uint32_t tb0, uint32_t k) {
    return k < AF_FIN_TB ? reinterpret_cast<const af_tb_t*>(tbs + k * (AF_TB_CIG + 1)) : tb + tb0 + k;
}
__device__ __forceinline__ const af_anchor_t* afw_anchor(const af_anchor_t* staged, const af_anchor_t* all, uint32_t j) { return j < AF_FIN_AN ? staged + j : all + j; }

#if defined(AF_CUTS)
#define AFW_CUT(bit) if (G.dbg & (bit)) { if (lane == 0) { moni_aln_rec_t rc_; memset(&rc_, 0, sizeof rc_); A.recs[r_in] = rc_; if (A.dev_len) { A.dev_len[r_in] = 0; A.dev_off[r_in] = 0; } } continue; }
#else
#define AFW_CUT(bit)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) finish_wave_kernel(const af_args_t G) {
    __shared__ af_finw_t L;
    const int lane = threadIdx.x;
    const ak_args_t& A = G.A;
    const ak_fmt_t& F = A.fmt;
    // the sequence names: in LDS for the kernel's lifetime when they fit
    const uint32_t n_seq = (uint32_t)A.P.n_seq;
    const bool names_lds = n_seq <= AFW_NSEQ && F.sname_off[n_seq] <= AFW_NAMES;
    if (names_lds) {
        for (uint32_t k = lane; k <= n_seq; k += 64) L.name_off[k] = (uint16_t)F.sname_off[k];
        for (uint32_t k = lane; k < F.sname_off[n_seq]; k += 64) L.names[k] = F.snames[k];
    }
    uint32_t* const tbs = reinterpret_cast<uint32_t*>(L.line);       // staged traceback records: the line buffer is free until the line is rendered
#define TB(k) (*afw_tb(tbs, G.tb, h_tb0, (k)))          // (the argument is evaluated once: call sites pass tbx++)
#define ANCH(j) (*afw_anchor(L.an, PL.an + h_an0, (j)))
#define NAME_LEN(sid) (names_lds ? (uint32_t)(L.name_off[(sid) + 1] - L.name_off[sid]) : F.sname_off[(sid) + 1] - F.sname_off[sid])
    // the plan's 40-byte header (status, final chain, strand, traced problems, window) says where everything else is: it is read with
    // one round trip, the next read's while this one is worked on
    uint64_t h0 = 0, h1 = 0, h2 = 0, h4 = 0, g0 = 0, g1 = 0, g2 = 0, g4 = 0;
    if (blockIdx.x < A.n_reads) { const uint64_t* H = reinterpret_cast<const uint64_t*>(&G.plans[blockIdx.x]); h0 = H[0]; h1 = H[1]; h2 = H[2]; h4 = H[4]; }
    for (uint64_t r_in = blockIdx.x; r_in < A.n_reads; r_in += gridDim.x, h0 = g0, h1 = g1, h2 = g2, h4 = g4) {
        if (r_in + gridDim.x < A.n_reads) { const uint64_t* H = reinterpret_cast<const uint64_t*>(&G.plans[r_in + gridDim.x]); g0 = H[0]; g1 = H[1]; g2 = H[2]; g4 = H[4]; }
        af_plan_t& PL = G.plans[r_in];
        const uint32_t st = (uint32_t)(h0 & 0xFFu);
        if (st == AF_ST_FALLBACK) continue;
        const uint32_t h_final = (uint32_t)(h1 & 0xFFu), h_nalt = (uint32_t)((h1 >> 8) & 0xFFu), h_strand = (uint32_t)((h1 >> 16) & 0xFFu), h_ntb = (uint32_t)((h1 >> 24) & 0xFFu);
        const int32_t h_score2 = (int32_t)(uint32_t)(h1 >> 32);
        const uint32_t h_tb0 = (uint32_t)h4, h_an0 = (uint32_t)(h4 >> 32) & 0xFFFFu, h_nan = (uint32_t)(h4 >> 48) & 0xFFu;
        const uint64_t r = A.read_lo + r_in;
        const uint64_t off = A.offs[r];
        const uint32_t m = (uint32_t)(A.offs[r + 1] - off);
        const bool aligned = st == AF_ST_FINAL;
        __syncthreads();
        AF_STAMP(fw0);
        // where the alignment starts in the lift tables: these dependent loads (directory, sequence record, first run) are started before the
        // staging below instead of inside lane 0's serial section
        const uint64_t aln_pos = aligned ? h2 : 0ull;
        uint32_t hint0 = 0xFFFFFFFFu;
        const uint32_t sid0 = aligned ? ac_seq_of(A.P, aln_pos, &hint0) : 0u;
        moni_lift_seq_t LS0; LS0.second = 0; LS0.run_off = 0; LS0.n_runs = 0; LS0.start = 0; LS0.end = 0;
        if (aligned) LS0 = A.P.lift_seqs[sid0];
        // the final chain's record, the alternatives and (below) its traceback records come with one round trip each instead of field by field
        if (aligned) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(&PL.cand[h_final]);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&L.cand);
            for (uint32_t w = lane; w < sizeof(af_cand_t) / 4; w += 64) dst[w] = src[w];
            const uint32_t* asrc = reinterpret_cast<const uint32_t*>(PL.an + h_an0);
            uint32_t* adst = reinterpret_cast<uint32_t*>(L.an);
            for (uint32_t w = lane; w < (h_nan < AF_FIN_AN ? h_nan : AF_FIN_AN) * (uint32_t)(sizeof(af_anchor_t) / 4); w += 64) adst[w] = asrc[w];
            if (lane < AF_MAX_CAND) L.alt_score[lane] = PL.alt_score[lane];
        }
        const af_cand_t* C = aligned ? &L.cand : nullptr;
        const uint32_t strand = aligned ? h_strand : 0u;
        const uint32_t n_alt = aligned ? h_nalt : 0u;
        if (aligned) {
            for (uint32_t w = lane; w < (h_ntb < AF_FIN_TB ? h_ntb : AF_FIN_TB) * AFW_TB_WORDS; w += 64) {
                const uint32_t k = w / AFW_TB_WORDS, x = w % AFW_TB_WORDS;
                tbs[w] = reinterpret_cast<const uint32_t*>(&G.tb[h_tb0 + k])[x];
            }
            if ((uint32_t)lane < n_alt) {       // the alternatives' sequences and 1-based positions: one lane each
                const uint64_t ap = PL.alt_pos[lane];
                const uint32_t s2 = ac_seq_of(A.P, ap);
                L.alt_sid[lane] = s2; L.alt_p1[lane] = (uint32_t)(ap - A.P.lift_seqs[s2].start + 1);
            }
        }
        // ---- the read in alignment orientation ----
        for (uint32_t k = lane; k < m && k < AF_MAX_READ; k += 64) L.seq[k] = strand ? ak_compl(A.D.reads[off + m - 1 - k]) : A.D.reads[off + k];
        __syncthreads();
        AFW_CUT(16384)         // timing experiment: stop after the staging
        bool ovf = false;
        uint64_t lifted = 0;
        if (aligned && lane == 0) {
            // ---- CIGAR stitching (aligner_ksw2.hpp:3049-3108) ----
            uint32_t n = 0;
#define PUSH(op) do { if (n < AFS_CIG) L.cig[n++] = (op); else ovf = true; } while (0)
#define PUSH_MERGE_FIRST(op, first) do { const uint32_t o_ = (op); if ((first) && (o_ & 0xf) == 0 && n > 0) L.cig[n - 1] += o_; else PUSH(o_); } while (0)
            uint32_t tbx = 0;
            if (C->overlap) {
                const af_tb_t& T = TB(tbx);
                if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH(T.ops[T.n_ops - 1 - k]);
            } else {
                if (C->has_lc) { const af_tb_t& T = TB(tbx++); if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH(T.ops[k]); }
                const uint32_t rc_x = C->has_rc ? tbx++ : 0u;
                for (uint32_t j = 0; j < C->n_an; ++j) {
                    const af_anchor_t g = ANCH(j);
                    const uint32_t mlen = g.len;
                    if (n > 0 && (L.cig[n - 1] & 0xf) == 0) L.cig[n - 1] += mlen << 4; else PUSH(mlen << 4);
                    if (j + 1 < C->n_an) {
                        if (g.gap_kind == AF_GAP_TASK) { const af_tb_t& T = TB(tbx++); if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH_MERGE_FIRST(T.ops[T.n_ops - 1 - k], k == 0); }
                        else if (g.gap_kind == AF_GAP_INS) PUSH_MERGE_FIRST(((uint32_t)(uint16_t)g.gap_val << 4) | 1u, true);
                        else if (g.gap_kind == AF_GAP_DEL0) PUSH_MERGE_FIRST(2u, true);
                        else if (g.gap_kind == AF_GAP_1X1) PUSH_MERGE_FIRST(1u << 4, true);
                    }
                }
                if (C->has_rc) { const af_tb_t& T = TB(rc_x); if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH_MERGE_FIRST(T.ops[T.n_ops - 1 - k], k == 0); }
            }
#undef PUSH
#undef PUSH_MERGE_FIRST
#if defined(AF_CUTS)
            if (G.dbg & 32768) { L.n_cig = n; L.n_lcig = 0; L.ovf = 1; } else {      // timing experiment: no lift
#endif
            // ---- the alignment lifted to the reference contig (aligner_ksw2.hpp:3133-3160) ----
            const moni_lift_seq_t LS = LS0;
            const moni_lift_run_t* __restrict__ runs = A.P.lift_runs + LS.run_off;
            const uint64_t start = aln_pos - LS.start;
            const uint32_t rel = hint0 == 0xFFFFFFFFu ? hint0 : hint0 - LS.run_off;
            uint64_t lp = 0;
            const int nl = ovf ? -1 : lift_cigar(runs, LS.n_runs, start, L.cig, n, L.lcig, AFS_LCIG, rel, &lp);
            if (nl < 0) ovf = true;
            lifted = LS.second + (ovf ? 0ull : lp);
            L.n_cig = n; L.n_lcig = nl < 0 ? 0u : (uint32_t)nl; L.ovf = ovf ? 1u : 0u;
#if defined(AF_CUTS)
            }
#endif
        }
        __syncthreads();
        AF_STAMP(fw1); AF_PROF(G, 16, fw0, fw1);
        AFW_CUT(2048)          // timing experiment: stop after staging, stitching, lifting
        if (aligned) { ovf = L.ovf != 0; lifted = ((uint64_t)(uint32_t)__shfl((int)(lifted >> 32), 0) << 32) | (uint32_t)__shfl((int)(lifted & 0xFFFFFFFFull), 0); }
        moni_aln_rec_t rec;
        rec.status = aligned ? 1u : 0u; rec.strand = strand; rec.ref_pos = aligned ? aln_pos : 0; rec.score = aligned ? C->score : 0; rec.score2 = aligned ? h_score2 : 0;
        rec.n_cigar = 0; rec.n_alt = 0; rec.cigar_off = 0; rec.alt_off = 0; rec.nm = 0; rec.md_len = 0; rec.md_off = 0; rec.txt_len = 0; rec.lift_nm = 0; rec.txt_off = 0;
        bool to_host = (aligned && ovf) || m > AF_MAX_READ;       // more CIGAR operations than the staging holds (align_kernel runs beside this kernel, its list is closed)
        if (aligned && ovf && lane == 0) atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CIGAR], 1u);
        uint32_t p = 0;
        const uint64_t n0 = F.rname_off[r], n1 = F.rname_off[r + 1];
        if (!to_host) {
            // ---- what the line needs besides the CIGARs: NM / MD of the lifted alignment, NM of the one on the pangenome text, MAPQ ----
            int nm = 0, lift_nm = 0, mapq = 0, oa_pos = 0, pos1 = 0;
            uint32_t n_md = 0, sid = 0, lsid = 0;
            bool mapped = false;
            const uint32_t n_cig = aligned ? L.n_cig : 0u, n_lcig = aligned ? L.n_lcig : 0u;
            const int32_t score = aligned ? C->score : 0, score2 = aligned ? h_score2 : 0;
            if (aligned) {
                uint64_t ref_len = 0;
                for (uint32_t k = 0; k < n_lcig; ++k) { const int op = L.lcig[k] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += L.lcig[k] >> 4; }
                mapped = ref_len > 0;
                sid = sid0; lsid = ac_seq_of(A.P, lifted);
                oa_pos = (int)(aln_pos - LS0.start + 1);
                pos1 = (int)(lifted - A.P.lift_seqs[lsid].start + 1);
                // compute_mapq_se_bwa (mapq.hpp:146-184), the operations in the host's order, none contracted
                {
                    const int32_t rl = mapped ? (int32_t)ref_len : 0;
                    const int32_t l = rl > (int32_t)m ? rl : (int32_t)m;
                    const int32_t sub = score2 ? score2 : F.min_len * F.smatch;
                    if (sub < score) {
                        const double identity = __dsub_rn(1., __ddiv_rn(__ddiv_rn((double)(l * F.smatch - score), (double)(F.smatch + F.smismatch)), (double)l));
                        if (score != 0) {
                            double tmp = (double)l < 50.0 ? 1. : ((uint32_t)l < F.mapq_tab_n ? F.mapq_tab[l] : F.mapq_tab[F.mapq_tab_n - 1]);
                            tmp = __dmul_rn(tmp, __dmul_rn(identity, identity));
                            const double v = __dadd_rn(__dmul_rn(__dmul_rn(__ddiv_rn(__dmul_rn(6.02, (double)(score - sub)), (double)F.smatch), tmp), tmp), .499);
                            mapq = (int)v;
                        }
                        if (mapq > 60) mapq = 60;
                        if (mapq < 0) mapq = 0;
                        mapq = (int)__dadd_rn(__dmul_rn((double)mapq, 1.), .499);
                    }
                }
                bool same = n_lcig == n_cig && lifted == aln_pos;
                for (uint32_t k = 0; same && k < n_cig; ++k) same = L.lcig[k] == L.cig[k];
                uint32_t dummy = 0;
                // the two reference windows (lifted, and on the pangenome text) go to LDS with one round trip: the MD walks read them there
                uint64_t ref_len_u = 0;
                for (uint32_t k = 0; k < n_cig; ++k) { const int op = L.cig[k] & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len_u += L.cig[k] >> 4; }
                const bool win = ref_len <= AFS_LINE / 2 && ref_len_u <= AFS_LINE / 2;
                if (win) {
                    if (mapped) for (uint32_t k = lane; k < (uint32_t)ref_len; k += 64) { const uint64_t a = lifted + k; L.line[k] = (uint8_t)dp_nt4(a < A.D.n_text ? A.D.text[a] : 0u); }
                    if (!same) for (uint32_t k = lane; k < (uint32_t)ref_len_u; k += 64) { const uint64_t a = aln_pos + k; L.line[AFS_LINE / 2 + k] = (uint8_t)dp_nt4(a < A.D.n_text ? A.D.text[a] : 0u); }
                    __syncthreads();
                }
                if (mapped) nm = afs_md(G.A.D, L, L.lcig, n_lcig, lifted, true, n_md, win ? L.line : nullptr);
                lift_nm = same ? nm : afs_md(G.A.D, L, L.cig, n_cig, aln_pos, false, dummy, win ? L.line + AFS_LINE / 2 : nullptr);
                if (n_md > AFS_MAXMD) { to_host = true; if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CIGAR], 1u); }
            }
            __syncthreads();
            AF_STAMP(fw2); AF_PROF(G, 17, fw1, fw2);
            AFW_CUT(4096)      // ... after MD / NM / MAPQ
            // ---- the segments of the line (sam.hpp:144-188), lane 0 ----
            if (lane == 0 && !to_host) {
                uint32_t n = 0, q = 0;
                afs_push(L, n, q, SK_RNAME, 0, (uint32_t)(n1 - n0));
                if (!aligned) {
                    afs_lits(L, n, q, LT_UNAL, 19);
                    L.seq_at = q; afs_push(L, n, q, SK_SEQ, 0, m); afs_lits(L, n, q, LT_TAB, 1);
                    L.qual_at = F.quals ? q : ~0u;
                    if (F.quals) afs_push(L, n, q, SK_QUAL, 0, m); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_NL, 1);
                } else {
                    afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, strand ? 16 : 0); afs_lits(L, n, q, LT_TAB, 1);
                    if (mapped) afs_push(L, n, q, SK_NAME, lsid, NAME_LEN(lsid)); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, mapped ? pos1 : 0); afs_lits(L, n, q, LT_TAB, 1); afs_num(L, n, q, mapq); afs_lits(L, n, q, LT_TAB, 1);
                    if (mapped) afs_cigar(L, n, q, L.lcig, n_lcig, L.lcig_off, 0u); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_MATE, 7);
                    L.seq_at = q; afs_push(L, n, q, SK_SEQ, 0, m); afs_lits(L, n, q, LT_TAB, 1);
                    L.qual_at = F.quals ? q : ~0u;
                    if (F.quals) afs_push(L, n, q, SK_QUAL, 0, m); else afs_lits(L, n, q, LT_STAR, 1);
                    afs_lits(L, n, q, LT_AS, 6); afs_num(L, n, q, score); afs_lits(L, n, q, LT_NM, 6); afs_num(L, n, q, mapped ? nm : 0);
                    if (score2 != 0) { afs_lits(L, n, q, LT_ZS, 6); afs_num(L, n, q, score2); }
                    afs_lits(L, n, q, LT_MD, 6);
                    {   // the MD string as one segment: where every item's text starts
                        uint32_t w = 0;
                        for (uint32_t k = 0; k < n_md; ++k) {
                            const uint32_t it = L.md_item[k], ty = it & 3u;
                            L.md_off[k] = (uint16_t)w;
                            w += afs_ndig((it >> 2) & 0x3FFu) + (ty == 1 ? 1u : ty == 2 ? 1u + ((it >> 12) & 0x1FFu) : 0u);
                        }
                        L.md_off[n_md] = (uint16_t)w;
                        afs_push(L, n, q, SK_MD, n_md, w);
                    }
                    afs_lits(L, n, q, LT_OA, 6); afs_push(L, n, q, SK_NAME, sid, NAME_LEN(sid));
                    afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, oa_pos); afs_lits(L, n, q, strand ? LT_MINUS : LT_PLUS, 3);
                    afs_cigar(L, n, q, L.cig, n_cig, L.cig_off, 1u);
                    afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, mapq); afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, lift_nm); afs_lits(L, n, q, LT_SEMI, 1);
                    afs_lits(L, n, q, LT_AA, 6);
                    for (uint32_t k = 0; k < n_alt; ++k) {
                        const uint32_t s2 = L.alt_sid[k];
                        afs_push(L, n, q, SK_NAME, s2, NAME_LEN(s2));
                        afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, (int)L.alt_p1[k]); afs_lits(L, n, q, LT_COMMA, 1); afs_num(L, n, q, L.alt_score[k]); afs_lits(L, n, q, LT_SEMI, 1);
                    }
                    afs_lits(L, n, q, LT_NL, 1);
                }
                if (n <= AFS_MAXSEG) L.seg_off[n] = (uint16_t)(q < 0xFFFFu ? q : 0xFFFFu);
                L.n_seg = n; L.total = q;
            }
            __syncthreads();
            AF_STAMP(fw3); AF_PROF(G, 18, fw2, fw3);
            AFW_CUT(8192)      // ... after the segment list
            const uint32_t n_seg = to_host ? 0u : L.n_seg;
            p = to_host ? 0u : L.total;
            if (!to_host && (n_seg > AFS_MAXSEG || p > AFS_LINE)) { to_host = true; if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CAPACITY], 1u); }
            if (!to_host) {
                // ---- SEQ and QUAL are plain copies; every lane renders its share of the other bytes of the line ----
                const uint32_t s_at = L.seq_at, q_at = L.qual_at, holes = m + (q_at != ~0u ? m : 0u);
                if (s_at + m <= AFS_LINE) for (uint32_t k = lane; k < m; k += 64) L.line[s_at + k] = L.seq[k];
                if (q_at != ~0u && q_at + m <= AFS_LINE) for (uint32_t k = lane; k < m; k += 64) L.line[q_at + k] = F.quals[strand ? off + m - 1 - k : off + k];
                for (uint32_t i = lane; i + holes < p; i += 64) {
                    uint32_t b = i;
                    if (b >= s_at) b += m;
                    if (q_at != ~0u && b >= q_at) b += m;
                    uint32_t lo = 0, hi = n_seg;                       // last segment that starts at or before b
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)L.seg_off[mid] <= b) lo = mid; else hi = mid; }
                    const uint32_t d = b - L.seg_off[lo], kind = L.seg_kind[lo], val = L.seg_val[lo];
                    uint8_t ch;
                    if (kind == SK_LIT) ch = (uint8_t)afs_lit[val + d];
                    else if (kind == SK_NUM || kind == SK_NEG) {
                        const uint32_t len = (uint32_t)L.seg_off[lo + 1] - L.seg_off[lo];
                        if (kind == SK_NEG && d == 0) ch = '-';
                        else { uint32_t u = val; for (uint32_t t = d + 1; t < len; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                    }
                    else if (kind == SK_SEQ) ch = L.seq[d];
                    else if (kind == SK_QUAL) ch = F.quals[strand ? off + m - 1 - d : off + d];
                    else if (kind == SK_NAME) ch = names_lds ? L.names[L.name_off[val] + d] : F.snames[F.sname_off[val] + d];
                    else if (kind == SK_RNAME) ch = F.rnames[n0 + d];
                    else if (kind == SK_CIG) {          // operation k of a CIGAR: its length, then its letter
                        const uint16_t* offs = (val & 1u) ? L.cig_off : L.lcig_off; const uint32_t* cg = (val & 1u) ? L.cig : L.lcig;
                        uint32_t a = 0, z = val >> 1;
                        while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)offs[mid] <= d) a = mid; else z = mid; }
                        const uint32_t e = d - offs[a], nd = (uint32_t)offs[a + 1] - offs[a] - 1u;
                        if (e == nd) ch = (uint8_t)afs_lit[LT_OPS + (cg[a] & 0xfu)];
                        else { uint32_t u = cg[a] >> 4; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                    } else {                            // SK_MD: item k of the MD string: the count of matches, then the base or ^ and the deleted bases
                        uint32_t a = 0, z = val;
                        while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)L.md_off[mid] <= d) a = mid; else z = mid; }
                        const uint32_t it = L.md_item[a], ty = it & 3u, run = (it >> 2) & 0x3FFu, e = d - L.md_off[a], nd = afs_ndig(run);
                        if (e < nd) { uint32_t u = run; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                        else if (ty == 1) { const uint32_t bc = (it >> 12) & 7u; ch = (uint8_t)afs_lit[LT_BASES + (bc > 4 ? 4 : bc)]; }
                        else if (e == nd) ch = '^';
                        else { const uint64_t ta = lifted + (it >> 21) + (e - nd - 1u); ch = (uint8_t)afs_lit[LT_BASES + dp_nt4(ta < A.D.n_text ? A.D.text[ta] : 0u)]; }
                    }
                    L.line[b] = ch;
                }
            }
            AF_STAMP(fw4); AF_PROF(G, 19, fw3, fw4);
        }
        __syncthreads();
        AF_STAMP(fw6);
        // ---- out: the text pool (8-byte words, bump-allocated), coalesced; the record ----
        if (to_host) rec.status = 2;           // does not fit the staging: the host pipeline redoes the read
        else {
            const unsigned long long words = (unsigned long long)((p + 7) >> 3);
            const uint32_t shard = blockIdx.x % AF_TXT_SHARDS;
            unsigned long long to = 0;
            if (lane == 0) to = atomicAdd(&G.txt_cur[shard * 8], words);
            to = ((unsigned long long)(uint32_t)__shfl((int)(to >> 32), 0) << 32) | (uint32_t)__shfl((int)(to & 0xFFFFFFFFull), 0);
            if (to + words > G.txt_shard_words) rec.status = 2;
            else {
                to += (unsigned long long)(shard + 1) * G.txt_shard_words;
                const uint64_t* src = reinterpret_cast<const uint64_t*>(L.line);
                for (unsigned long long k = lane; k < words; k += 64) F.txt_pool[to + k] = src[k];
                rec.txt_len = p; rec.txt_off = to;
            }
        }
        if (lane == 0) {
            A.recs[r_in] = rec;
            if (A.dev_len) {
                A.dev_len[r_in] = rec.txt_len; A.dev_off[r_in] = rec.txt_off;
                if (rec.status == 2 || rec.txt_len == 0) atomicAdd(&A.dev_sum[0], 1ull);
                else if (rec.status == 1) atomicAdd(&A.dev_sum[8 + 8 * (blockIdx.x % 16)], 1ull);      // sharded: one address takes only ~50 M atomics/s
            }
        }
        AF_STAMP(fw7); AF_PROF(G, 21, fw6, fw7); AF_PROF(G, 22, fw0, fw7);
    }
#undef TB
#undef ANCH
#undef NAME_LEN
}


// ------------------------------------------------------------------------------------------------------------------------------
// finish_prep_kernel + finish_render_kernel: finish_wave_kernel's work in two kernels.  What lane 0 of a wavefront did alone there - stitch the CIGAR
// (aligner_ksw2.hpp:3049-3108), lift it (:3133-3160), MAPQ (mapq.hpp:146-184), and the MD / NM walks (sam.hpp:249-287), which were 64 columns wide but a
// few items long - is ONE LANE's work per read here: 64 reads per wavefront, 1 / 64 of the wavefront instructions (finish_wave_kernel issued ~3500 per
// read, 63 of 64 lanes idle in two thirds of them: profiles/r04t).  The lane leaves a "recipe" in HBM (AFP_WORDS words per read: header numbers, the
// alternatives, the stitched and the lifted CIGAR, the MD items).  finish_render_kernel, one wavefront per read, stages the recipe, lays the line's segments
// out BY ALL LANES (a segment's kind and length follow from its index; offsets by a wavefront scan - finish_wave_kernel's lane 0 listed them one by one) and
// renders and stores the bytes as before.  Same bytes, same records, same hand-overs to the host pipeline.
// ------------------------------------------------------------------------------------------------------------------------------
#define AFP_WORDS 512u
#define AFP_ALT 16u              // 3 words per alternative: sequence, 1-based position, score
#define AFP_CIG 64u
#define AFP_LCIG 128u
#define AFP_MD 256u
enum { AFP_F_ALIGNED = 1u, AFP_F_MAPPED = 2u, AFP_F_STRAND = 4u, AFP_F_HOST = 8u };
enum { AFP_H_FLAGS = 0, AFP_H_NCIG, AFP_H_NMD, AFP_H_NM, AFP_H_LIFTNM, AFP_H_MAPQ, AFP_H_SCORE, AFP_H_SCORE2, AFP_H_POS1, AFP_H_OAPOS, AFP_H_SIDS, AFP_H_LIFTED_LO, AFP_H_LIFTED_HI,
       AFP_H_POS_LO, AFP_H_POS_HI, AFP_H_N };
static_assert(AFP_H_N <= AFP_ALT && AFP_ALT + 3 * AF_MAX_CAND <= AFP_CIG && AFP_CIG + AFS_CIG <= AFP_LCIG && AFP_LCIG + AFS_LCIG <= AFP_MD && AFP_MD + AFS_MAXMD <= AFP_WORDS, "recipe layout");

// a sequence of bytes read front to back by one lane, eight per load (af_bytes_t; element e = byte start + e, or start - e)
struct afl_stream_t { af_bytes_t S; uint64_t w; int g; };
__device__ __forceinline__ afl_stream_t afl_stream(const uint8_t* base, uint64_t start, uint64_t limit, bool rev) { afl_stream_t x; x.S = af_bytes(base, start, limit, rev); x.w = 0; x.g = -1; return x; }
__device__ __forceinline__ uint32_t afl_at(afl_stream_t& x, uint32_t e) {
    const int g = (int)(e >> 3);
    if (g != x.g) { x.w = af_group(x.S, g); x.g = g; }
    return (uint32_t)(x.w >> (8 * (e & 7u))) & 0xFFu;
}
// The 2-bit forms of the read (the seeding stage's pattern workspace: code and mask words of the read's strand-resolved pattern, 64 words apart) and of the text
struct afl_two_t { const uint64_t* pat; uint64_t cb, mb, n2w; const uint64_t* text2; uint64_t n_tw; const uint32_t* exc; uint32_t exc_sh; };
// 32 bases from base `first` of a 2-bit sequence whose words lie `stride` apart (n_w of them)
__device__ __forceinline__ uint64_t afl_bits2(const uint64_t* __restrict__ w, uint64_t stride, uint64_t n_w, uint64_t first) {
    const uint64_t i = first >> 5; const uint32_t sh = 2u * (uint32_t)(first & 31u);
    const uint64_t lo = i < n_w ? w[i * stride] : 0ull;
    if (!sh) return lo;
    const uint64_t hi = i + 1 < n_w ? w[(i + 1) * stride] : 0ull;
    return (lo >> sh) | (hi << (64u - sh));
}
// MD / NM of one CIGAR (write_MD_core), one lane: afs_md's items, one after the other.  A stretch of matches / mismatches is compared 32 columns at a time through the
// 2-bit forms (one XOR; an item per set bit) - column by column only where a byte outside A / C / G / T may lie (the read's mask words, the text's exception
// bitmap): with 64 reads per wavefront the column loop's ~100 instructions per column were this kernel's whole cost (profiles/r04v).
__device__ __forceinline__ int afl_md(const dp_launch_t& D, const afl_two_t& W, uint64_t off, uint32_t m, uint32_t strand, const uint32_t* __restrict__ cg, uint32_t n_cig, uint64_t t0,
                                      bool items, uint32_t& n_items, uint32_t* __restrict__ md) {
    afl_stream_t TS = afl_stream(D.text, t0, D.text_limit, false), RS = afl_stream(D.reads, strand ? off + m - 1 : off, D.reads_limit, strand != 0);
    int NM = 0; uint32_t l_MD = 0, ni = 0;
    bool bad = false;
    uint64_t t = t0; uint32_t q = 0;
    for (uint32_t i = 0; i < n_cig; ++i) {
        const uint32_t op = cg[i] & 0xf, len = cg[i] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (uint32_t k0 = 0; k0 < len; k0 += 32) {
                const uint32_t n = len - k0 < 32 ? len - k0 : 32;
                const uint64_t ta = t + k0; const uint32_t qa = q + k0;
                const uint64_t lowm = n < 32 ? (1ull << (2 * n)) - 1ull : ~0ull;
                bool two = W.pat != nullptr && ta + n <= D.n_text && qa + n <= m;
                uint64_t pw = 0, tw = 0;
                if (two) {
                    const uint64_t b0 = ta >> W.exc_sh, b1 = (ta + n - 1) >> W.exc_sh;
                    if (((W.exc[b0 >> 5] >> (b0 & 31u)) | (W.exc[b1 >> 5] >> (b1 & 31u))) & 1u) two = false;
                    else if (afl_bits2(W.pat + W.mb, 64, W.n2w, qa) & lowm) two = false;
                    else { pw = afl_bits2(W.pat + W.cb, 64, W.n2w, qa); tw = afl_bits2(W.text2, 1, W.n_tw, ta); }
                }
                if (two) {
                    uint64_t x = pw ^ tw;
                    x = (x | (x >> 1)) & 0x5555555555555555ull & lowm;
                    uint32_t prev = 0;
                    while (x) {
                        const uint32_t b = (uint32_t)__builtin_ctzll(x), k = b >> 1, c2 = (uint32_t)(tw >> b) & 3u;
                        const uint32_t tc = c2 == 2 ? 3u : c2 == 3 ? 2u : c2;          // (b >> 1) & 3: A 0, C 1, T 2, G 3 -> seq_nt4_table's A 0, C 1, G 2, T 3
                        const uint32_t run = l_MD + (k - prev);
                        if (items) { if (ni < AFS_MAXMD) md[ni] = 1u | ((run & 0x3FFu) << 2) | (tc << 12); if (run > 0x3FFu) bad = true; }
                        ++ni; ++NM; l_MD = 0; prev = k + 1;
                        x &= x - 1;
                    }
                    l_MD += n - prev;
                } else {
                    for (uint32_t k = k0; k < k0 + n; ++k) {
                        const uint64_t a = t + k;
                        const uint32_t tc = a < D.n_text ? dp_nt4(afl_at(TS, (uint32_t)(a - t0))) : dp_nt4(0u);
                        const uint32_t rb = q + k < m ? afl_at(RS, q + k) : 0u;
                        const uint32_t rc = dp_nt4(strand ? (uint32_t)ak_compl((uint8_t)rb) : rb);          // the read in alignment orientation (kpbseq.h:120-137), as finish_wave_kernel stages it
                        if (rc != tc) {
                            if (items) { if (ni < AFS_MAXMD) md[ni] = 1u | ((l_MD & 0x3FFu) << 2) | (tc << 12); if (l_MD > 0x3FFu) bad = true; }
                            ++ni; ++NM; l_MD = 0;
                        } else ++l_MD;
                    }
                }
            }
            q += len; t += len;
        } else if (op == 1) { q += len; NM += (int)len; }
        else if (op == 2) {
            if (items) {
                if (ni < AFS_MAXMD) md[ni] = 2u | ((l_MD & 0x3FFu) << 2) | ((len & 0x1FFu) << 12) | ((uint32_t)(t - t0) << 21);
                if (l_MD > 0x3FFu || len > 0x1FFu || (t - t0) > 0x7FFu) bad = true;
            }
            ++ni;
            l_MD = 0; t += len; NM += (int)len;
        } else if (op == 3) t += len;
    }
    if (items && l_MD > 0) { if (ni < AFS_MAXMD) md[ni] = (l_MD & 0x3FFu) << 2; if (l_MD > 0x3FFu) bad = true; ++ni; }
    n_items = (bad || ni > AFS_MAXMD) ? AFS_MAXMD + 1 : ni;
    return NM;
}

__global__ void __launch_bounds__(64) finish_prep_kernel(const af_args_t G, uint32_t* __restrict__ recipes) {
    const ak_args_t& A = G.A;
    const ak_fmt_t& F = A.fmt;
    for (uint64_t r_in = (uint64_t)blockIdx.x * 64 + threadIdx.x; r_in < A.n_reads; r_in += (uint64_t)gridDim.x * 64) {
        const af_plan_t& PL = G.plans[r_in];
        const uint64_t* H = reinterpret_cast<const uint64_t*>(&PL);
        const uint64_t h0 = H[0], h1 = H[1], h2 = H[2], h4 = H[4];
        const uint32_t st = (uint32_t)(h0 & 0xFFu);
        if (st == AF_ST_FALLBACK) continue;
        uint32_t* __restrict__ S = recipes + (size_t)r_in * AFP_WORDS;
        const bool aligned = st == AF_ST_FINAL;
        if (!aligned) { S[AFP_H_FLAGS] = 0; continue; }
        const uint32_t h_final = (uint32_t)(h1 & 0xFFu), n_alt = (uint32_t)((h1 >> 8) & 0xFFu), strand = (uint32_t)((h1 >> 16) & 0xFFu);
        const int32_t score2 = (int32_t)(uint32_t)(h1 >> 32);
        const uint32_t h_tb0 = (uint32_t)h4, h_an0 = (uint32_t)(h4 >> 32) & 0xFFFFu;
        const uint64_t r = A.read_lo + r_in;
        const uint64_t off = A.offs[r];
        const uint32_t m = (uint32_t)(A.offs[r + 1] - off);
        const uint64_t aln_pos = h2;
        uint32_t hint0 = 0xFFFFFFFFu;
        const uint32_t sid0 = ac_seq_of(A.P, aln_pos, &hint0);
        const moni_lift_seq_t LS0 = A.P.lift_seqs[sid0];
        const af_cand_t C = PL.cand[h_final];
        uint32_t* const cig = S + AFP_CIG; uint32_t* const lcig = S + AFP_LCIG;
        // ---- CIGAR stitching (aligner_ksw2.hpp:3049-3108); the last operation written is kept in a register (a match run merges into it) ----
        uint32_t n = 0, last = 0; bool ovf = false;
#define PUSH(op) do { if (n < AFS_CIG) { last = (op); cig[n++] = last; } else ovf = true; } while (0)
#define PUSH_MERGE_FIRST(op, first) do { const uint32_t o_ = (op); if ((first) && (o_ & 0xf) == 0 && n > 0) { last += o_; cig[n - 1] = last; } else PUSH(o_); } while (0)
        uint32_t tbx = h_tb0;
        if (C.overlap) {
            const af_tb_t& T = G.tb[tbx];
            if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH(T.ops[T.n_ops - 1 - k]);
        } else {
            if (C.has_lc) { const af_tb_t& T = G.tb[tbx++]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH(T.ops[k]); }
            const uint32_t rc_x = C.has_rc ? tbx++ : 0u;
            const af_anchor_t* const AN = PL.an + h_an0;
            for (uint32_t j = 0; j < C.n_an; ++j) {
                const af_anchor_t g = AN[j];
                const uint32_t mlen = g.len;
                if (n > 0 && (last & 0xf) == 0) { last += mlen << 4; cig[n - 1] = last; } else PUSH(mlen << 4);
                if (j + 1 < C.n_an) {
                    if (g.gap_kind == AF_GAP_TASK) { const af_tb_t& T = G.tb[tbx++]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH_MERGE_FIRST(T.ops[T.n_ops - 1 - k], k == 0); }
                    else if (g.gap_kind == AF_GAP_INS) PUSH_MERGE_FIRST(((uint32_t)(uint16_t)g.gap_val << 4) | 1u, true);
                    else if (g.gap_kind == AF_GAP_DEL0) PUSH_MERGE_FIRST(2u, true);
                    else if (g.gap_kind == AF_GAP_1X1) PUSH_MERGE_FIRST(1u << 4, true);
                }
            }
            if (C.has_rc) { const af_tb_t& T = G.tb[rc_x]; if (T.n_ops == ~0u) ovf = true; else for (uint32_t k = 0; k < T.n_ops; ++k) PUSH_MERGE_FIRST(T.ops[T.n_ops - 1 - k], k == 0); }
        }
#undef PUSH
#undef PUSH_MERGE_FIRST
#if defined(AF_CUTS)
        if (G.dbg & 0x100000u) { S[AFP_H_FLAGS] = n; continue; }          // timing experiments (results are wrong): stop after the stitching ...
#endif
        // ---- the alignment lifted to the reference contig (aligner_ksw2.hpp:3133-3160) ----
        const moni_lift_run_t* __restrict__ runs = A.P.lift_runs + LS0.run_off;
        const uint32_t rel = hint0 == 0xFFFFFFFFu ? hint0 : hint0 - LS0.run_off;
        uint64_t lp = 0;
        const int nl = ovf ? -1 : lift_cigar(runs, LS0.n_runs, aln_pos - LS0.start, cig, n, lcig, AFS_LCIG, rel, &lp);
        if (nl < 0) ovf = true;
        const uint64_t lifted = LS0.second + (ovf ? 0ull : lp);
        const uint32_t n_cig = n, n_lcig = nl < 0 ? 0u : (uint32_t)nl;
#if defined(AF_CUTS)
        if (G.dbg & 0x200000u) { S[AFP_H_FLAGS] = n_lcig + (uint32_t)lifted; continue; }          // ... after the lift
#endif
        uint32_t flags = AFP_F_ALIGNED | (strand ? AFP_F_STRAND : 0u);
        int nm = 0, lift_nm = 0, mapq = 0, oa_pos = 0, pos1 = 0;
        uint32_t n_md = 0, lsid = 0;
        if (ovf || m > AF_MAX_READ) flags |= AFP_F_HOST;      // more operations than the staging holds: the host pipeline redoes the read
        else {
            uint64_t ref_len = 0;
            for (uint32_t k = 0; k < n_lcig; ++k) { const uint32_t o = lcig[k]; const int op = o & 0xf; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += o >> 4; }
            const bool mapped = ref_len > 0;
            if (mapped) flags |= AFP_F_MAPPED;
            lsid = ac_seq_of(A.P, lifted);
            oa_pos = (int)(aln_pos - LS0.start + 1);
            pos1 = (int)(lifted - A.P.lift_seqs[lsid].start + 1);
            const int32_t score = C.score;
            {   // compute_mapq_se_bwa (mapq.hpp:146-184), the operations in the host's order, none contracted
                const int32_t rl = mapped ? (int32_t)ref_len : 0;
                const int32_t l = rl > (int32_t)m ? rl : (int32_t)m;
                const int32_t sub = score2 ? score2 : F.min_len * F.smatch;
                if (sub < score) {
                    const double identity = __dsub_rn(1., __ddiv_rn(__ddiv_rn((double)(l * F.smatch - score), (double)(F.smatch + F.smismatch)), (double)l));
                    if (score != 0) {
                        double tmp = (double)l < 50.0 ? 1. : ((uint32_t)l < F.mapq_tab_n ? F.mapq_tab[l] : F.mapq_tab[F.mapq_tab_n - 1]);
                        tmp = __dmul_rn(tmp, __dmul_rn(identity, identity));
                        const double v = __dadd_rn(__dmul_rn(__dmul_rn(__ddiv_rn(__dmul_rn(6.02, (double)(score - sub)), (double)F.smatch), tmp), tmp), .499);
                        mapq = (int)v;
                    }
                    if (mapq > 60) mapq = 60;
                    if (mapq < 0) mapq = 0;
                    mapq = (int)__dadd_rn(__dmul_rn((double)mapq, 1.), .499);
                }
            }
            bool same = n_lcig == n_cig && lifted == aln_pos;
            for (uint32_t k = 0; same && k < n_cig; ++k) same = lcig[k] == cig[k];
            uint32_t dummy = 0;
            afl_two_t W;
            W.pat = (G.pat && G.text2 && G.exc && !G.pe) ? G.pat : nullptr; W.text2 = G.text2; W.n_tw = (A.D.n_text + 31) >> 5; W.exc = G.exc; W.exc_sh = G.exc_sh; W.cb = W.mb = W.n2w = 0;
            if (W.pat) {
                const uint64_t task = 2 * r + strand, pb = ws_pat_base(G.blk, task), lb = ws_block_len(G.blk, task);
                W.cb = pb + 64u * ((lb + 7) / 8); W.n2w = (lb + 31) / 32; W.mb = W.cb + 64u * W.n2w;
            }
#if defined(AF_CUTS)
            if (G.dbg & 0x400000u) { S[AFP_H_FLAGS] = (uint32_t)mapq + lsid; continue; }          // ... in front of the MD walks
#endif
            if (mapped) nm = afl_md(A.D, W, off, m, strand, lcig, n_lcig, lifted, true, n_md, S + AFP_MD);
            lift_nm = same ? nm : afl_md(A.D, W, off, m, strand, cig, n_cig, aln_pos, false, dummy, nullptr);
            if (n_md > AFS_MAXMD) { flags |= AFP_F_HOST; n_md = 0; }
            for (uint32_t k = 0; k < n_alt && k < AF_MAX_CAND; ++k) {       // the alternatives' sequences and 1-based positions
                const uint64_t ap = PL.alt_pos[k];
                const uint32_t s2 = ac_seq_of(A.P, ap);
                S[AFP_ALT + 3 * k] = s2; S[AFP_ALT + 3 * k + 1] = (uint32_t)(ap - A.P.lift_seqs[s2].start + 1); S[AFP_ALT + 3 * k + 2] = (uint32_t)PL.alt_score[k];
            }
        }
        S[AFP_H_FLAGS] = flags | (n_alt << 8); S[AFP_H_NCIG] = n_cig | (n_lcig << 16); S[AFP_H_NMD] = n_md; S[AFP_H_NM] = (uint32_t)nm; S[AFP_H_LIFTNM] = (uint32_t)lift_nm;
        S[AFP_H_MAPQ] = (uint32_t)mapq; S[AFP_H_SCORE] = (uint32_t)C.score; S[AFP_H_SCORE2] = (uint32_t)score2; S[AFP_H_POS1] = (uint32_t)pos1; S[AFP_H_OAPOS] = (uint32_t)oa_pos;
        S[AFP_H_SIDS] = sid0 | (lsid << 16); S[AFP_H_LIFTED_LO] = (uint32_t)lifted; S[AFP_H_LIFTED_HI] = (uint32_t)(lifted >> 32); S[AFP_H_POS_LO] = (uint32_t)aln_pos; S[AFP_H_POS_HI] = (uint32_t)(aln_pos >> 32);
    }
}

#define AFR_MAXSEG 136u          // 35 fixed segments + 6 per alternative + the newline
struct af_finr_t {
    uint8_t line[AFS_LINE];
    uint8_t seq[AF_MAX_READ];
    uint32_t hdr[AFP_CIG];               // the recipe's header and alternatives
    uint32_t cig[AFS_CIG], lcig[AFS_LCIG];
    uint32_t md_item[AFS_MAXMD];
    uint16_t md_off[AFS_MAXMD + 1], cig_off[AFS_CIG + 1], lcig_off[AFS_LCIG + 1];
    uint8_t names[AFW_NAMES]; uint16_t name_off[AFW_NSEQ + 2];
    uint16_t seg_off[AFR_MAXSEG + 1]; uint8_t seg_kind[AFR_MAXSEG]; uint32_t seg_val[AFR_MAXSEG];
};
// exclusive prefix sums of len[0 .. n) over the wavefront, 64 at a time: off[k] = sum of the lengths before k, off[n] = the total (returned).  len(k) is evaluated by lane k & 63.
template <class LenF>
__device__ __forceinline__ uint32_t afr_scan(uint32_t n, uint16_t* off, LenF len) {
    const int lane = threadIdx.x;
    uint32_t base = 0;
    for (uint32_t k0 = 0; k0 < n; k0 += 64) {
        const uint32_t k = k0 + lane;
        const uint32_t mine = k < n ? len(k) : 0u;
        uint32_t inc = mine;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t x = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += x; }
        if (k < n) off[k] = (uint16_t)(base + inc - mine < 0xFFFFu ? base + inc - mine : 0xFFFFu);
        base += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) off[n] = (uint16_t)(base < 0xFFFFu ? base : 0xFFFFu);
    return base;
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) finish_render_kernel(const af_args_t G, const uint32_t* __restrict__ recipes) {
    __shared__ af_finr_t L;
    const int lane = threadIdx.x;
    const ak_args_t& A = G.A;
    const ak_fmt_t& F = A.fmt;
    const uint32_t n_seq = (uint32_t)A.P.n_seq;
    const bool names_lds = n_seq <= AFW_NSEQ && F.sname_off[n_seq] <= AFW_NAMES;
    if (names_lds) {
        for (uint32_t k = lane; k <= n_seq; k += 64) L.name_off[k] = (uint16_t)F.sname_off[k];
        for (uint32_t k = lane; k < F.sname_off[n_seq]; k += 64) L.names[k] = F.snames[k];
    }
#define NAME_LEN(sid) (names_lds ? (uint32_t)(L.name_off[(sid) + 1] - L.name_off[sid]) : F.sname_off[(sid) + 1] - F.sname_off[sid])
    // a read's status and the recipe's first words come with one round trip, the next read's while this one is rendered
    uint32_t st_n = AF_ST_FALLBACK, hw_n = 0;
    if (blockIdx.x < A.n_reads) { st_n = G.plans[blockIdx.x].status; hw_n = recipes[(size_t)blockIdx.x * AFP_WORDS + lane]; }
    for (uint64_t r_in = blockIdx.x; r_in < A.n_reads; r_in += gridDim.x) {
        const uint32_t st = st_n, hw = hw_n;
        if (r_in + gridDim.x < A.n_reads) { st_n = G.plans[r_in + gridDim.x].status; hw_n = recipes[(size_t)(r_in + gridDim.x) * AFP_WORDS + lane]; }
        if (st == AF_ST_FALLBACK) continue;
        const uint32_t* __restrict__ S = recipes + (size_t)r_in * AFP_WORDS;
        __syncthreads();
        L.hdr[lane] = hw;
        __syncthreads();
        const uint32_t flags = L.hdr[AFP_H_FLAGS];
        const bool aligned = (flags & AFP_F_ALIGNED) != 0 && st == AF_ST_FINAL;
        const bool mapped = (flags & AFP_F_MAPPED) != 0;
        const uint32_t strand = aligned && (flags & AFP_F_STRAND) ? 1u : 0u;
        const uint32_t n_alt = aligned ? (flags >> 8) & 0xFFu : 0u;
        const uint32_t n_cig = aligned ? L.hdr[AFP_H_NCIG] & 0xFFFFu : 0u, n_lcig = aligned ? L.hdr[AFP_H_NCIG] >> 16 : 0u, n_md = aligned ? L.hdr[AFP_H_NMD] : 0u;
        const uint64_t r = A.read_lo + r_in;
        const uint64_t off = A.offs[r];
        const uint32_t m = (uint32_t)(A.offs[r + 1] - off);
        const uint64_t lifted = (uint64_t)L.hdr[AFP_H_LIFTED_LO] | ((uint64_t)L.hdr[AFP_H_LIFTED_HI] << 32);
        const uint64_t aln_pos = aligned ? (uint64_t)L.hdr[AFP_H_POS_LO] | ((uint64_t)L.hdr[AFP_H_POS_HI] << 32) : 0ull;
        const int32_t score = aligned ? (int32_t)L.hdr[AFP_H_SCORE] : 0, score2 = aligned ? (int32_t)L.hdr[AFP_H_SCORE2] : 0;
        bool to_host = (aligned && (flags & AFP_F_HOST)) || m > AF_MAX_READ;
        if (aligned && (flags & AFP_F_HOST) && lane == 0) atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CIGAR], 1u);
        // ---- the CIGARs, the MD items and the read in alignment orientation ----
        if (!to_host) {
            for (uint32_t k = lane; k < n_cig; k += 64) L.cig[k] = S[AFP_CIG + k];
            for (uint32_t k = lane; k < n_lcig; k += 64) L.lcig[k] = S[AFP_LCIG + k];
            for (uint32_t k = lane; k < n_md; k += 64) L.md_item[k] = S[AFP_MD + k];
            for (uint32_t k = lane; k < m; k += 64) L.seq[k] = strand ? ak_compl(A.D.reads[off + m - 1 - k]) : A.D.reads[off + k];
        }
        __syncthreads();
        moni_aln_rec_t rec;
        rec.status = aligned ? 1u : 0u; rec.strand = strand; rec.ref_pos = aln_pos; rec.score = score; rec.score2 = score2;
        rec.n_cigar = 0; rec.n_alt = 0; rec.cigar_off = 0; rec.alt_off = 0; rec.nm = 0; rec.md_len = 0; rec.md_off = 0; rec.txt_len = 0; rec.lift_nm = 0; rec.txt_off = 0;
        uint32_t p = 0;
        const uint64_t n0 = F.rname_off[r], n1 = F.rname_off[r + 1];
        if (!to_host) {
            const int nm = (int)L.hdr[AFP_H_NM], lift_nm = (int)L.hdr[AFP_H_LIFTNM], mapq = (int)L.hdr[AFP_H_MAPQ], pos1 = (int)L.hdr[AFP_H_POS1], oa_pos = (int)L.hdr[AFP_H_OAPOS];
            const uint32_t sid = L.hdr[AFP_H_SIDS] & 0xFFFFu, lsid = L.hdr[AFP_H_SIDS] >> 16;
            // ---- where the texts of the CIGAR operations and of the MD items start ----
            uint32_t w_lcig = 0, w_cig = 0, w_md = 0;
            if (aligned) {
                w_lcig = afr_scan(n_lcig, L.lcig_off, [&](uint32_t k) { return afs_ndig(L.lcig[k] >> 4) + 1u; });
                w_cig = afr_scan(n_cig, L.cig_off, [&](uint32_t k) { return afs_ndig(L.cig[k] >> 4) + 1u; });
                w_md = afr_scan(n_md, L.md_off, [&](uint32_t k) { const uint32_t it = L.md_item[k], ty = it & 3u; return afs_ndig((it >> 2) & 0x3FFu) + (ty == 1 ? 1u : ty == 2 ? 1u + ((it >> 12) & 0x1FFu) : 0u); });
            }
            // ---- the segments of the line (sam.hpp:144-188): segment i's kind, value and length from i alone, its offset by a scan ----
            const uint32_t n_seg = aligned ? 36u + 6u * n_alt : 6u;
            const uint32_t has_q = F.quals ? 1u : 0u;
            auto seg = [&](uint32_t i, uint32_t& kind, uint32_t& val) -> uint32_t {
                auto lit = [&](uint32_t at, uint32_t len) { kind = SK_LIT; val = at; return len; };
                auto num = [&](int v) { if (v < 0) { const uint32_t u = 0u - (uint32_t)v; kind = SK_NEG; val = u; return afs_ndig(u) + 1u; } kind = SK_NUM; val = (uint32_t)v; return afs_ndig((uint32_t)v); };
                if (!aligned) {
                    switch (i) {
                    case 0: kind = SK_RNAME; val = 0; return (uint32_t)(n1 - n0);
                    case 1: return lit(LT_UNAL, 19);
                    case 2: kind = SK_SEQ; val = 0; return m;
                    case 3: return lit(LT_TAB, 1);
                    case 4: if (has_q) { kind = SK_QUAL; val = 0; return m; } return lit(LT_STAR, 1);
                    default: return lit(LT_NL, 1);
                    }
                }
                if (i >= 35u && i < 35u + 6u * n_alt) {
                    const uint32_t k = (i - 35u) / 6u, x = (i - 35u) % 6u;
                    const uint32_t s2 = L.hdr[AFP_ALT + 3 * k];
                    switch (x) {
                    case 0: kind = SK_NAME; val = s2; return NAME_LEN(s2);
                    case 1: return lit(LT_COMMA, 1);
                    case 2: return num((int)L.hdr[AFP_ALT + 3 * k + 1]);
                    case 3: return lit(LT_COMMA, 1);
                    case 4: return num((int)L.hdr[AFP_ALT + 3 * k + 2]);
                    default: return lit(LT_SEMI, 1);
                    }
                }
                if (i >= 35u) return lit(LT_NL, 1);
                switch (i) {
                case 0: kind = SK_RNAME; val = 0; return (uint32_t)(n1 - n0);
                case 1: return lit(LT_TAB, 1);
                case 2: return num(strand ? 16 : 0);
                case 3: return lit(LT_TAB, 1);
                case 4: if (mapped) { kind = SK_NAME; val = lsid; return NAME_LEN(lsid); } return lit(LT_STAR, 1);
                case 5: return lit(LT_TAB, 1);
                case 6: return num(mapped ? pos1 : 0);
                case 7: return lit(LT_TAB, 1);
                case 8: return num(mapq);
                case 9: return lit(LT_TAB, 1);
                case 10: if (mapped) { kind = SK_CIG; val = 0u | (n_lcig << 1); return w_lcig; } return lit(LT_STAR, 1);
                case 11: return lit(LT_MATE, 7);
                case 12: kind = SK_SEQ; val = 0; return m;
                case 13: return lit(LT_TAB, 1);
                case 14: if (has_q) { kind = SK_QUAL; val = 0; return m; } return lit(LT_STAR, 1);
                case 15: return lit(LT_AS, 6);
                case 16: return num(score);
                case 17: return lit(LT_NM, 6);
                case 18: return num(mapped ? nm : 0);
                case 19: return score2 != 0 ? lit(LT_ZS, 6) : lit(LT_ZS, 0);
                case 20: if (score2 != 0) return num(score2); return lit(LT_ZS, 0);
                case 21: return lit(LT_MD, 6);
                case 22: kind = SK_MD; val = n_md; return w_md;
                case 23: return lit(LT_OA, 6);
                case 24: kind = SK_NAME; val = sid; return NAME_LEN(sid);
                case 25: return lit(LT_COMMA, 1);
                case 26: return num(oa_pos);
                case 27: return lit(strand ? LT_MINUS : LT_PLUS, 3);
                case 28: kind = SK_CIG; val = 1u | (n_cig << 1); return w_cig;
                case 29: return lit(LT_COMMA, 1);
                case 30: return num(mapq);
                case 31: return lit(LT_COMMA, 1);
                case 32: return num(lift_nm);
                case 33: return lit(LT_SEMI, 1);
                default: return lit(LT_AA, 6);          // 34
                }
            };
            p = afr_scan(n_seg, L.seg_off, [&](uint32_t i) { uint32_t kind = 0, val = 0; const uint32_t len = seg(i, kind, val); L.seg_kind[i] = (uint8_t)kind; L.seg_val[i] = val; return len; });
            __syncthreads();
            if (p > AFS_LINE) { to_host = true; if (lane == 0) atomicAdd(&G.ctr[AFC_WHY + AF_WHY_CAPACITY], 1u); }
            if (!to_host) {
                // ---- SEQ and QUAL are plain copies; every lane renders its share of the other bytes of the line ----
                const uint32_t s_at = L.seg_off[aligned ? 12 : 2], q_at = has_q ? (uint32_t)L.seg_off[aligned ? 14 : 4] : ~0u, holes = m + (q_at != ~0u ? m : 0u);
                if (s_at + m <= AFS_LINE) for (uint32_t k = lane; k < m; k += 64) L.line[s_at + k] = L.seq[k];
                if (q_at != ~0u && q_at + m <= AFS_LINE) for (uint32_t k = lane; k < m; k += 64) L.line[q_at + k] = F.quals[strand ? off + m - 1 - k : off + k];
                for (uint32_t i = lane; i + holes < p; i += 64) {
                    uint32_t b = i;
                    if (b >= s_at) b += m;
                    if (q_at != ~0u && b >= q_at) b += m;
                    uint32_t lo = 0, hi = n_seg;                       // last segment that starts at or before b (an empty segment shares its offset with the next: never the last such)
                    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)L.seg_off[mid] <= b) lo = mid; else hi = mid; }
                    const uint32_t d = b - L.seg_off[lo], kind = L.seg_kind[lo], val = L.seg_val[lo];
                    uint8_t ch;
                    if (kind == SK_LIT) ch = (uint8_t)afs_lit[val + d];
                    else if (kind == SK_NUM || kind == SK_NEG) {
                        const uint32_t len = (uint32_t)L.seg_off[lo + 1] - L.seg_off[lo];
                        if (kind == SK_NEG && d == 0) ch = '-';
                        else { uint32_t u = val; for (uint32_t t = d + 1; t < len; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                    }
                    else if (kind == SK_SEQ) ch = L.seq[d];
                    else if (kind == SK_QUAL) ch = F.quals[strand ? off + m - 1 - d : off + d];
                    else if (kind == SK_NAME) ch = names_lds ? L.names[L.name_off[val] + d] : F.snames[F.sname_off[val] + d];
                    else if (kind == SK_RNAME) ch = F.rnames[n0 + d];
                    else if (kind == SK_CIG) {          // operation k of a CIGAR: its length, then its letter
                        const uint16_t* offs = (val & 1u) ? L.cig_off : L.lcig_off; const uint32_t* cg = (val & 1u) ? L.cig : L.lcig;
                        uint32_t a = 0, z = val >> 1;
                        while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)offs[mid] <= d) a = mid; else z = mid; }
                        const uint32_t e = d - offs[a], nd = (uint32_t)offs[a + 1] - offs[a] - 1u;
                        if (e == nd) ch = (uint8_t)afs_lit[LT_OPS + (cg[a] & 0xfu)];
                        else { uint32_t u = cg[a] >> 4; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                    } else {                            // SK_MD: item k of the MD string: the count of matches, then the base or ^ and the deleted bases
                        uint32_t a = 0, z = val;
                        while (z - a > 1) { const uint32_t mid = (a + z) >> 1; if ((uint32_t)L.md_off[mid] <= d) a = mid; else z = mid; }
                        const uint32_t it = L.md_item[a], ty = it & 3u, run = (it >> 2) & 0x3FFu, e = d - L.md_off[a], nd = afs_ndig(run);
                        if (e < nd) { uint32_t u = run; for (uint32_t t = e + 1; t < nd; ++t) u /= 10u; ch = (uint8_t)('0' + u % 10u); }
                        else if (ty == 1) { const uint32_t bc = (it >> 12) & 7u; ch = (uint8_t)afs_lit[LT_BASES + (bc > 4 ? 4 : bc)]; }
                        else if (e == nd) ch = '^';
                        else { const uint64_t ta = lifted + (it >> 21) + (e - nd - 1u); ch = (uint8_t)afs_lit[LT_BASES + dp_nt4(ta < A.D.n_text ? A.D.text[ta] : 0u)]; }
                    }
                    L.line[b] = ch;
                }
            }
        }
        __syncthreads();
        // ---- out: the text pool (8-byte words, bump-allocated), coalesced; the record ----
        if (to_host) rec.status = 2;           // does not fit the staging: the host pipeline redoes the read
        else {
            const unsigned long long words = (unsigned long long)((p + 7) >> 3);
            const uint32_t shard = blockIdx.x % AF_TXT_SHARDS;
            unsigned long long to = 0;
            if (lane == 0) to = atomicAdd(&G.txt_cur[shard * 8], words);
            to = ((unsigned long long)(uint32_t)__shfl((int)(to >> 32), 0) << 32) | (uint32_t)__shfl((int)(to & 0xFFFFFFFFull), 0);
            if (to + words > G.txt_shard_words) rec.status = 2;
            else {
                to += (unsigned long long)(shard + 1) * G.txt_shard_words;
                const uint64_t* src = reinterpret_cast<const uint64_t*>(L.line);
                for (unsigned long long k = lane; k < words; k += 64) F.txt_pool[to + k] = src[k];
                rec.txt_len = p; rec.txt_off = to;
            }
        }
        if (lane == 0) {
            A.recs[r_in] = rec;
            if (A.dev_len) {
                A.dev_len[r_in] = rec.txt_len; A.dev_off[r_in] = rec.txt_off;
                if (rec.status == 2 || rec.txt_len == 0) atomicAdd(&A.dev_sum[0], 1ull);
                else if (rec.status == 1) atomicAdd(&A.dev_sum[8 + 8 * (blockIdx.x % 16)], 1ull);      // sharded: one address takes only ~50 M atomics/s
            }
        }
    }
#undef NAME_LEN
}

// ------------------------------------------------------------------------------------------------------------------------------
// gather_lines_kernel: the SAM lines of a sub-batch, which the kernels wrote to the text pool in completion order, copied into one
// block in read order (pos = exclusive scan of the line lengths): the host receives the block with one transfer and does nothing else
// ------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_lines_kernel(const uint64_t* __restrict__ pool, const uint64_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                           const uint64_t* __restrict__ pos, uint64_t n_reads, uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t l = len[r];
        const uint8_t* __restrict__ src = reinterpret_cast<const uint8_t*>(pool + off[r]);
        uint8_t* __restrict__ dst = out + pos[r];
        // destination offsets are byte-granular: a head up to the first 8-byte boundary, aligned words, a tail
        const uint64_t head = l < 8 ? l : ((8 - (reinterpret_cast<uintptr_t>(dst) & 7)) & 7);
        if ((uint64_t)lane < head) dst[lane] = src[lane];
        const uint64_t words = (l - head) >> 3;
        for (uint64_t k = lane; k < words; k += 64) {
            uint64_t w;
            memcpy(&w, src + head + 8 * k, 8);
            *reinterpret_cast<uint64_t*>(dst + head + 8 * k) = w;
        }
        const uint64_t done = head + 8 * words;
        if (done + lane < l) dst[done + lane] = src[done + lane];
    }
}
// the per-sub-batch summary the host reads: [0] bytes of the block, [1] records that need the host, [2] aligned reads
__global__ void gather_summary_kernel(const uint64_t* __restrict__ len, const uint64_t* __restrict__ pos, uint64_t n_reads, const unsigned long long* __restrict__ dev_sum,
                                      unsigned long long* __restrict__ out3) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out3[0] = n_reads ? pos[n_reads - 1] + len[n_reads - 1] : 0ull;
        out3[1] = dev_sum[0];
        unsigned long long al = dev_sum[1];
        for (int s = 0; s < 16; ++s) al += dev_sum[8 + 8 * s];
        out3[2] = al;
    }
}

// The DP stage of the staged kernels, single-end and paired alike: the reads' (pairs') problems have been queued by bin_tasks_kernel; extension and gap problems
// by tile, then the global problems: their records (global_task_kernel), their bands and queues (global_band_kernel), banded where a narrow band is proven,
// the full matrix otherwise.  n_units: plans of the launch (reads or pairs).
// chunks up to which dp_wave_kernel takes a group (a pair costs it ~0.1 ms of one SIMD; dp_lane_kernel's launch ~2 ms whatever the count): MONI_AF_WAVE_MAX, 0 = never
static inline uint32_t af_wave_max() { const char* v = getenv("MONI_AF_WAVE_MAX"); return v ? (uint32_t)strtoul(v, nullptr, 10) : 96u; }
static inline void af_launch_dp(const af_args_t& G, hipStream_t sx, unsigned dp_grid, unsigned n_cu, uint64_t n_units) {
    if (!(G.dbg & 0x20000u)) hipLaunchKernelGGL(band_tasks_kernel, dim3((unsigned)((G.bin_cap + 255) / 256)), dim3(256), 0, sx, G);      // (the provisional list holds at most bin_cap problems)
    hipLaunchKernelGGL(af_chunk_kernel, dim3(1), dim3(64), 0, sx, G, (uint32_t)AF_GRP_LARGE, (uint32_t)AF_GRP_WILD);
    hipLaunchKernelGGL((dp_lane_kernel<AF_BLK, AF_QCAP, AF_LPASS>), dim3(dp_grid), dim3(64), 0, sx, G, (uint32_t)AF_GRP_LARGE);
    hipLaunchKernelGGL((dp_lane_kernel<AF_TS, AF_TS, 1>), dim3(dp_grid), dim3(64), 0, sx, G, (uint32_t)AF_GRP_SMALL);
    hipLaunchKernelGGL((dp_band_kernel<AF_BANDW>), dim3(n_cu * 8), dim3(64), 0, sx, G, (uint32_t)AF_GRP_BAND);
    hipLaunchKernelGGL((dp_lane_kernel<AF_BLK, AF_QCAP, AF_LPASS, true>), dim3(dp_grid), dim3(64), 0, sx, G, (uint32_t)AF_GRP_WILD);
    hipLaunchKernelGGL((dp_wave_kernel<AF_BLK, AF_LPASS, 4>), dim3(n_cu * 8), dim3(64), 0, sx, G, (uint32_t)AF_GRP_WILD, (uint32_t)AF_GRP_WILD);
    hipLaunchKernelGGL(global_task_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, sx, G);          // (a global problem's window comes from the extensions of its chain: all of them are through)
    const uint64_t gmax = G.task_cap > n_units * AF_MAX_TASKS_READ ? G.task_cap - n_units * AF_MAX_TASKS_READ : 0;          // global problems the slots hold
    if (gmax) hipLaunchKernelGGL(global_band_kernel, dim3((unsigned)((gmax + 255) / 256)), dim3(256), 0, sx, G);
    hipLaunchKernelGGL(af_chunk_kernel, dim3(1), dim3(64), 0, sx, G, (uint32_t)AF_GRP_GLOBAL, (uint32_t)AF_GRP_GWILD);
    hipLaunchKernelGGL((dp_band_kernel<AF_BANDW>), dim3(n_cu * 8), dim3(64), 0, sx, G, (uint32_t)AF_GRP_GBAND);
    hipLaunchKernelGGL((dp_lane_kernel<AF_GBLK, AF_QCAP, AF_GPASS>), dim3(dp_grid), dim3(64), 0, sx, G, (uint32_t)AF_GRP_GLOBAL);
    hipLaunchKernelGGL((dp_lane_kernel<AF_GBLK, AF_QCAP, AF_GPASS, true>), dim3(dp_grid), dim3(64), 0, sx, G, (uint32_t)AF_GRP_GWILD);
    hipLaunchKernelGGL((dp_wave_kernel<AF_GBLK, AF_GPASS, 8>), dim3(n_cu * 8), dim3(64), 0, sx, G, (uint32_t)AF_GRP_GLOBAL, (uint32_t)AF_GRP_GWILD);
}
