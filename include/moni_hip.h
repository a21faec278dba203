/*
 * moni_hip.h — C ABI of libmoni_hip.so, the MI355X (gfx950) implementation of the moni-align
 * per-read hot path.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * The reference has no FFI layer of its own: its seams are C++ template
 * concepts and one C function from ksw2.  Each entry point below names the reference interface
 * it stands in for (paths relative to the reference checkout).
 *
 * Conventions: opaque handles; every function returns 0 on success and a negative MONI_E* code
 * on failure and never calls exit(); the caller owns every input buffer; the library owns all
 * device memory; one moni_ctx_t per host thread / HIP stream (not thread-safe per ctx, thread-safe
 * across ctxs — mirrors include/aligner/align_reads_dispatcher.hpp:226-235: shared const index,
 * per-thread everything else).
 *
 * What runs where: every kernel-side stage of the path runs on the GPU only - without a HIP device moni_index_create returns
 * MONI_ENODEV, and nothing under oracle/ is linked or called.  The library does contain host code that is part of the product,
 * not a fallback for a missing GPU: (1) the "host pipeline" (align_host.hpp, pe_host.hpp, pe_big.cpp) redoes, with DP batches on
 * the GPU, the few reads or pairs whose seeds / anchors / chains / CIGAR exceed even the general kernel's capacities (0 reads per
 * 1 M in the benchmark; the statistics report them as handed_back), and (2) moni_align_csv_batch (`-c`) sends EVERY read through
 * that host pipeline by design, because the per-read MEM statistics are taken inside its chaining loop.
 */
#ifndef MONI_HIP_H
#define MONI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MONI_OK 0
#define MONI_EINVAL (-22)
#define MONI_ENOMEM (-12)
#define MONI_EIO (-5)
#define MONI_ENODEV (-19)   /* no HIP device / kernel launch failed */
#define MONI_ERANGE (-34)   /* index violates an invariant the device layout relies on */

typedef struct moni_index moni_index_t;   /* device-resident index image (one per GPU) */
typedef struct moni_ctx moni_ctx_t;       /* stream + workspaces + resident read batch */

/* Semantic content of <prefix>.thrbv.full.lcp.ms + .plain.slp + .ldx as flat host arrays
 * (field order of include/aligner/moni_lcp.hpp:208-225; values as built by
 * include/ms/moni.hpp:148-251, include/ms/ms_rle_string.hpp:245-303,
 * include/ms/thresholds_ds.hpp:393-430, include/common/seqidx.hpp:215-238). */
typedef struct {
    uint64_t n;              /* bwt.size() = text length + 1 */
    uint64_t r;              /* number of BWT runs */
    uint64_t w;              /* separator width */
    uint64_t n_seq;          /* number of sequences */
    const uint64_t *F;       /* [256] */
    const uint8_t *heads;    /* [r]   run heads (bytes <= 1 stored as 1) */
    const uint64_t *starts;  /* [r+1] run start positions, starts[r] = n */
    const uint64_t *ssa;     /* [r]   samples_start */
    const uint64_t *esa;     /* [r]   samples_last */
    const uint64_t *thr;     /* [r]   threshold position of each run, 0 = first run of its letter */
    const uint64_t *slcp;    /* [r]   LCP at each run start */
    const uint8_t *text;     /* [n-1] */
    const uint64_t *seq_starts; /* [n_seq+1] onsets in text coordinates */
    const char *seq_names;   /* n_seq NUL-terminated names back to back (seqidx names), or NULL */
    /* liftidx::lifts (include/aligner/liftidx.hpp:131-143), one lift::Lift per sequence, or all NULL for a FASTA-built
     * index (null lifts, liftidx.hpp:150-157): lift_second[i] = start of the target contig in the concatenation,
     * lift_len[i] = alignment columns, sorted positions of the ones of the levioSAM ins / del bit-vectors (ragged). */
    const uint64_t *lift_second;  /* [n_seq] */
    const uint64_t *lift_len;     /* [n_seq] */
    const uint64_t *lift_ins_off; /* [n_seq+1] */
    const uint64_t *lift_ins;
    const uint64_t *lift_del_off; /* [n_seq+1] */
    const uint64_t *lift_del;
} moni_flat_index_t;

/* Ragged batch of reads: read i is seq[offsets[i] .. offsets[i+1]).  Replaces the kseq_t batches of
 * include/common/kpbseq.h:315-326 (kbseq_read). */
typedef struct {
    const uint8_t *seq;
    const uint64_t *offsets; /* [n_reads+1] */
    uint64_t n_reads;
} moni_read_batch_t;

/* One MEM / seed, fields of include/aligner/mems.hpp:31-60 (count_dict is internal). */
typedef struct {
    uint64_t pos;          /* position in the text */
    uint32_t len;
    uint32_t idx;          /* position in the read */
    uint32_t rpos;
    uint32_t mate;         /* MATE_1|MATE_F = 0, MATE_1|MATE_RC = 2 */
    uint32_t total_occ;
    uint32_t num_filtered;
    uint64_t occ_off;      /* into the occs array */
    uint32_t occ_cnt;
    uint32_t read;         /* read index in the batch */
} moni_mem_t;

/* Seeding parameters: seed_finder ctor (include/aligner/seed_finder.hpp:64-68). */
typedef struct {
    uint32_t min_len;      /* -l, default 25 */
    uint32_t filter_seeds; /* -f off, default on */
    uint32_t n_seeds_thr;  /* -S, wrapper default 1000 */
    uint32_t report_mems;  /* populate_seeds(mems, report_mems) */
} moni_seed_params_t;

/* ksw_extz_t result fields (lh3/ksw2 ksw2.h) of one DP problem. */
typedef struct {
    int32_t max, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, zdropped;
    uint32_t n_cigar;
    uint32_t cigar_off;    /* into the cigar pool */
} moni_dp_result_t;

/* One ksw_extz2_sse call: query = qseq[q_off .. q_off+qlen), target = tseq[t_off .. t_off+tlen), nt4 codes. */
typedef struct {
    uint64_t q_off, t_off;
    int32_t qlen, tlen;
    int32_t flag;          /* KSW_EZ_* */
    int32_t reserved;
} moni_dp_task_t;

typedef struct {
    int8_t m;              /* 5 */
    int8_t mat[25];        /* ksw_gen_simple_mat, include/aligner/aligner_ksw2.hpp:3199-3211 */
    int8_t q, e;           /* gapo 4, gape 2 */
    int32_t w, zdrop, end_bonus; /* -1, -1, 400 (aligner_ksw2.hpp:110-113) */
} moni_dp_params_t;

/* ---- index ------------------------------------------------------------------------------- */
/* Replaces seed_finder's loading of .thrbv.full.lcp.ms/.plain.slp/.ldx (seed_finder.hpp:64-124):
 * converts the semantic arrays to the device layout and uploads it to `device`.  flat->text may be NULL: the text is redundant with
 * the r-index and is then rebuilt on the device by inverting the BWT (LF walks from the 2 r sampled positions, each down to the next
 * smaller sample), and checked against the BWT's symbol counts. */
int moni_index_create(const moni_flat_index_t *flat, int device, moni_index_t **out);
/* Same, from a MONIFLT2 file written by moni_align_amd/index_build.py. */
int moni_index_load(const char *path, int device, moni_index_t **out);
void moni_index_destroy(moni_index_t *idx);
uint64_t moni_index_n(const moni_index_t *idx);
uint64_t moni_index_r(const moni_index_t *idx);
uint64_t moni_index_device_bytes(const moni_index_t *idx);
/* The text the index was built over (n - 1 bytes: what PlainSlp::expandSubstr(0, n - 1) returns, seed_finder.hpp:88-99), whether it was
 * handed over or rebuilt from the BWT. */
int moni_index_text(const moni_index_t *idx, uint8_t *out, uint64_t cap);

/* ---- context / resident batch ------------------------------------------------------------ */
int moni_ctx_create(moni_index_t *idx, moni_ctx_t **out);
void moni_ctx_destroy(moni_ctx_t *ctx);
/* Copy a read batch to HBM (replaces rc_copy_kseq_t + kseq storage; the reverse-complement strand is
 * derived on the device with the table of include/common/kpbseq.h:120-137). */
int moni_reads_upload(moni_ctx_t *ctx, const moni_read_batch_t *batch);
/* Exchange the resident batch with the one parked in `slot` (0..255; an unused slot holds an empty batch): a caller that cycles
 * through several batches (one rank's shard of a read set, align_reads_dispatcher.hpp:300-345 reads them one after the other) uploads
 * each once, parks it, and swaps it in before moni_align_run.  Pointer exchange only; no copy, no kernel. */
int moni_reads_swap(moni_ctx_t *ctx, uint32_t slot);

/* ---- matching statistics: ms_t::query (include/ms/moni.hpp:292-295, 568-624) -------------- */
/* Device-only run over the resident batch, both strands (aligner_ksw2.hpp:333-334). */
int moni_ms_run(moni_ctx_t *ctx);
/* Host-buffer form: pointers[2*offsets[i] + s*len_i + k] = pointer k of strand s (0 fwd, 1 rc) of read i. */
int moni_ms_query_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, uint64_t *pointers);

/* Legacy `moni ms` / `moni mems` (src/matching_statistics.cpp:236-278, src/mems.cpp:236-280): pointers and matching-statistics
 * lengths of every read as given (forward strand), pointers[offsets[i] - offsets[0] + k] / lengths[...] for read offset k. */
int moni_ms_lengths_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, uint64_t *pointers, uint64_t *lengths);

/* ---- seeds: seed_finder::find_mems + populate_seeds (seed_finder.hpp:126-166, 258-318) ----- */
/* Device-only run (ms + mems + occurrences) over the resident batch. */
int moni_seed_run(moni_ctx_t *ctx, const moni_seed_params_t *prm);
/* Sizes of the last moni_seed_run. */
int moni_seed_counts(moni_ctx_t *ctx, uint64_t *n_mems, uint64_t *n_occs);
/* Copy the last result to host: mems in the order of the reference's per-read `mems` vector
 * (forward MEMs, reverse-complement MEMs, then for every MEM in that order its two halves),
 * read_mem_off[n_reads+1] delimits reads. */
int moni_seed_fetch(moni_ctx_t *ctx, moni_mem_t *mems, uint64_t *occs, uint64_t *read_mem_off);
/* Host-buffer form of the three calls above; *mems / *occs / *read_mem_off are malloc'ed, free with moni_free. */
int moni_seed_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const moni_seed_params_t *prm,
                    moni_mem_t **mems, uint64_t *n_mems, uint64_t **occs, uint64_t *n_occs, uint64_t **read_mem_off);
void moni_free(void *p);

/* ---- phi: moni_lcp::Phi_lcp / Phi_inv_lcp (include/aligner/moni_lcp.hpp:230-272) ---------- */
int moni_phi_lcp_batch(moni_ctx_t *ctx, const uint64_t *pos, uint64_t n, int inverse, uint64_t *out_pos, uint64_t *out_lcp);

/* ---- seed extension: ksw_extz2_sse (thirdparty/ksw2; call sites aligner_ksw2.hpp:2812-3015) */
int moni_extz_batch(moni_ctx_t *ctx, const moni_dp_params_t *prm, const uint8_t *qseq, uint64_t qseq_len,
                    const uint8_t *tseq, uint64_t tseq_len, const moni_dp_task_t *tasks, uint64_t n_tasks,
                    moni_dp_result_t *results, uint32_t *cigar_pool, uint64_t cigar_pool_cap, uint64_t *cigar_pool_used);

/* ---- the whole single-end path: aligner::align (include/aligner/aligner_ksw2.hpp:314-521) over a batch --- */
/* aligner::config_t (aligner_ksw2.hpp:84-130) with the `moni align` wrapper defaults (pipeline/moni.in:748-768). */
typedef struct {
    uint32_t min_len, ext_len, check_k, region_dist;      /* 25, 100, 5, 10 */
    uint32_t filter_seeds, n_seeds_thr, filter_freq, left_mem_check; /* 1, 1000, 1, 1 */
    double freq_thr;                                      /* 0.5 */
    int8_t smatch, smismatch, gapo, gapo2, gape, gape2;   /* 2, 4, 4, 13, 2, 1 */
    int32_t end_bonus, w, zdrop;                          /* 400, -1, -1 */
    int64_t max_dist_x, max_dist_y, max_iter, max_pred, min_chain_score, min_chain_length; /* 500,100,10,5,40,1 */
    uint32_t host_threads;                                /* threads for the host stages (chaining, stitching, SAM) */
    uint32_t reserved;
} moni_align_params_t;

typedef struct {
    uint64_t reads, aligned, dp_tasks, dp_cells, dp_rounds;
    double t_seed, t_chain, t_dp, t_host;                 /* seconds: seeding incl. fetch, chaining, DP batches incl. transfers, other host work */
    double t_dp_kernel;                                   /* seconds inside the align / extz kernels (HIP events) */
    uint64_t handed_back;                                 /* reads that exceeded the align kernel's capacities and went through the host pipeline */
    uint64_t dp_reused, dp_cells_reused;                  /* DP problems (and their cells) answered from the per-read memo of identical problems; not in dp_tasks/dp_cells */
    uint64_t kernel_fallback;                             /* reads outside the staged kernels' common case, taken by the general align kernel */
    uint64_t dp_ref_bytes;                                /* text bytes of the DP targets (the R of SURVEY.md 8(d)) */
    double t_k_chain, t_k_dp, t_k_select, t_k_finish;     /* HIP-event seconds of the staged kernels by group, summed over the sub-batches (launches of two
                                                             streams overlap: the sum exceeds the span t_dp_kernel) */
    uint64_t handover_why[12];                            /* why reads left the staged kernels (their sum can exceed kernel_fallback + handed_back: a read is counted once per
                                                             reason met): 0 read of 512 bases or more, 1 seeds / anchors beyond the largest LDS instance, 2 chains, 3 chains to
                                                             score, 4 anchors of the chains to score, 5 a DP problem beyond the register tile, 6 (unused), 7 wildcard base or
                                                             direction-bit budget, 8 selection loop depends on a score, 9 extension short of the query end, 10 a queue or pool
                                                             full, 11 CIGAR / MD / line beyond the staging (host pipeline) */
    uint64_t dp_cells_cut;                                /* staged DP kernels: cells of the problems after an extension's target rows that cannot hold its result are
                                                             cut (dp_cells counts the problems as the reference poses them: qlen x tlen of every ksw_extz2_sse call) */
    uint64_t dp_slots;                                    /* ... and the cell slots those kernels ran (128 problems x the chunk's longest query x the target rows of its passes) */
} moni_align_stats_t;

void moni_align_params_default(moni_align_params_t *p);
/* Replaces the per-read loop of st_align/mt_align (include/aligner/align_reads_dispatcher.hpp:346-357): SAM records of
 * the batch in input order, no header.  names: ragged bytes with name_off[n_reads+1]; quals: same offsets as the reads
 * or NULL.  *sam is malloc'ed (moni_free). */
int moni_align_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                     const uint8_t *quals, const moni_align_params_t *prm, char **sam, uint64_t *sam_len,
                     moni_align_stats_t *stats);
/* The same over the batch that moni_reads_upload made resident (reads already in HBM when the call starts).  *sam points
 * into a buffer the context owns: valid until the next moni_align_run / moni_ctx_destroy on this context, NOT to be freed
 * (a streaming caller writes it out and calls again; the pages stay mapped between batches). */
int moni_align_run(moni_ctx_t *ctx, const uint8_t *names, const uint64_t *name_off, const uint8_t *quals,
                   const moni_align_params_t *prm, char **sam, uint64_t *sam_len, moni_align_stats_t *stats);
/* moni_align_batch for a streaming caller: reads in host memory (uploaded by the call, no host copy kept), *sam in the context-owned
 * pinned buffer of moni_align_run (valid until the next moni_align_run / moni_align_stream / moni_ctx_destroy on this context, NOT to
 * be freed): the lines arrive in read order by one DMA per sub-batch, no host thread touches the text. */
int moni_align_stream(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                      const uint8_t *quals, const moni_align_params_t *prm, char **sam, uint64_t *sam_len,
                      moni_align_stats_t *stats);
/* aligner::align with csv (-c; aligner_ksw2.hpp:340-343, 417, include/common/csv.hpp:26-67): the SAM records and, per read, one line
 * `name,unique MEMs,total occurrences,max frequency,min frequency,highest / lowest count on one genome,filtered,chains skipped` (no header:
 * aligner::to_csv, aligner_ksw2.hpp:3230-3234).  A diagnostics mode: every read takes the host pipeline over the GPU's seeds and DP batches
 * (the selection loop counts the chains it skips); both texts are malloc'ed (moni_free). */
int moni_align_csv_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                         const uint8_t *quals, const moni_align_params_t *prm, char **sam, uint64_t *sam_len,
                         char **csv, uint64_t *csv_len, moni_align_stats_t *stats);
/* aligner::align with report_mems (-m; aligner_ksw2.hpp:346-373): one secondary record per MEM occurrence.  *sam is malloc'ed. */
int moni_report_mems_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                           const uint8_t *quals, const moni_align_params_t *prm, char **sam, uint64_t *sam_len);
/* ---- the paired-end path: aligner::align(kpbseq_t*) (aligner_ksw2.hpp:888-918, 1000-1326, 1536-1640) -------------------------------- */
/* The batch holds the pairs interleaved: reads 2p and 2p+1 are mate 1 and mate 2 of pair p (kpbseq_t's two kbseq_t,
 * include/common/kpbseq.h:300-326).  find_orphan: orphan recovery (aligner_ksw2.hpp:1536-1640, 2329-2720) for the pairs that chain but fail
 * jointly; its local alignment is klib's ksw_align (an absent submodule) restated as plain DP with its tie rules. */
typedef struct {
    uint32_t filter_dir, find_orphan;                     /* 1, 1: aligner::config_t::filter_dir, find_orphan (aligner_ksw2.hpp:113,128) */
    double dir_thr;                                       /* 50.0 */
    uint64_t ins_learning_n;                              /* 1000 */
    uint64_t ins_learning_score_gap_threshold;            /* 0 */
    uint32_t secondary_chains, reserved;                  /* 0: -Z, find_chains_secondary instead of find_chains (include/aligner/chain.hpp:442-727, aligner_ksw2.hpp:1190-1191);
                                                           * the staged paired kernels keep the second track of the chaining in LDS (pe_plan_kernel's SEC instances) */
} moni_pe_params_t;
/* The insert-size model (aligner_ksw2.hpp:3252-3262): zero-initialise, feed batches to moni_pe_learn_batch until complete != 0 (or
 * the input ends), then align - the order of st_align's paired loop (align_reads_dispatcher.hpp:356-389). */
typedef struct {
    double mean, std_dev, variance, sample_variance, m2;
    uint64_t count;
    uint32_t complete, reserved;
} moni_pe_model_t;
void moni_pe_params_default(moni_pe_params_t *p);
/* aligner::learn_fragment_model (aligner_ksw2.hpp:816-885) over one batch: updates *model.  MONI_ERANGE: a pair exceeded the
 * kernel's capacities and those of the host pipeline for pairs behind it (32 k anchors, mates of 32 k bases). */
int moni_pe_learn_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const moni_align_params_t *prm,
                        const moni_pe_params_t *pe, moni_pe_model_t *model);
/* The two SAM records of every pair, in input order, no header.  names / name_off / quals as in moni_align_batch (2N reads).
 * *sam is malloc'ed (moni_free). */
int moni_pe_align_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                        const uint8_t *quals, const moni_align_params_t *prm, const moni_pe_params_t *pe,
                        const moni_pe_model_t *model, char **sam, uint64_t *sam_len, moni_align_stats_t *stats);
/* The same for a streaming caller: *sam points into the pinned text buffer the context owns (moni_align_run's: valid until the next
 * moni_pe_align_stream / moni_pe_align_batch / moni_align_run / moni_align_stream / moni_ctx_destroy on this context, NOT to be freed).  The
 * two lines of every pair are written by the GPU and arrive in input order by one transfer per chunk of pairs; moni_pe_align_batch is this
 * call plus a copy into a malloc'ed block. */
int moni_pe_align_stream(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                        const uint8_t *quals, const moni_align_params_t *prm, const moni_pe_params_t *pe,
                        const moni_pe_model_t *model, char **sam, uint64_t *sam_len, moni_align_stats_t *stats);
/* The same over the interleaved pairs that moni_reads_upload made resident (reads already in HBM when the call starts; names and qualities
 * are host buffers as in moni_align_run); *sam in the context's buffer as for moni_pe_align_stream. */
int moni_pe_align_run(moni_ctx_t *ctx, const uint8_t *names, const uint64_t *name_off, const uint8_t *quals, const moni_align_params_t *prm,
                      const moni_pe_params_t *pe, const moni_pe_model_t *model, char **sam, uint64_t *sam_len, moni_align_stats_t *stats);
/* aligner::align(kpbseq_t*, out, csv_out) with csv (-c for pairs; aligner_ksw2.hpp:888-918, 1030-1031, 1066-1075, 1115-1118, 1354-1358;
 * include/common/csv.hpp:26-67): the SAM records of moni_pe_align_batch and one line of MEM statistics per PAIR under mate 1's name (the MEMs of a
 * pair are kept together: alignment.record_csv writes csv_m1 alone, aligner_ksw2.hpp:787-791).  A diagnostics mode like moni_align_csv_batch: the
 * counts of filtered MEMs and skipped chains exist only in the selection loop, so every pair takes the host's state machine (pe_big.cpp) over the GPU's
 * seeds and DP batches.  *sam and *csv are malloc'ed (moni_free). */
int moni_pe_align_csv_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                            const uint8_t *quals, const moni_align_params_t *prm, const moni_pe_params_t *pe,
                            const moni_pe_model_t *model, char **sam, uint64_t *sam_len, char **csv, uint64_t *csv_len,
                            moni_align_stats_t *stats);
/* aligner::align(paired_alignment_t&) with report_mems (-m for pairs; aligner_ksw2.hpp:1118-1180): one secondary record per occurrence of every MEM
 * the direction and frequency filters leave, under its mate's name.  *sam is malloc'ed. */
int moni_pe_report_mems_batch(moni_ctx_t *ctx, const moni_read_batch_t *batch, const uint8_t *names, const uint64_t *name_off,
                              const uint8_t *quals, const moni_align_params_t *prm, const moni_pe_params_t *pe, char **sam, uint64_t *sam_len);
/* aligner::to_sam (aligner_ksw2.hpp:3213-3219): "@HD", one "@SQ" per sequence, "@PG". */
int moni_sam_header(const moni_index_t *idx, char **sam, uint64_t *sam_len);

/* ---- the reference's on-disk liftidx (<prefix>.ldx: include/aligner/liftidx.hpp:117-143 over include/common/seqidx.hpp:197-238) ---- */
/* Both layouts load: the current one (u64 w after u) and the older one the reference's fixture data/Chr21.10.ldx has. */
int moni_ldx_info(const char *path, uint64_t *n_seq, uint64_t *u, uint64_t *w, int *has_w);
/* Load and write again (with_w: current layout).  Writing is sdsl-exact: the fixture round-trips byte for byte. */
int moni_ldx_rewrite(const char *in_path, const char *out_path, int with_w);
/* liftidx::serialize of a flat index's sequences and lifts (null lifts, liftidx.hpp:150-157, when idx carries none). */
int moni_ldx_write(const moni_flat_index_t *idx, const char *path, int with_w);
/* liftidx::lift (liftidx.hpp:89-95) of n text positions with the lifts of an .ldx file, on the GPU (lift_core.h tables). */
int moni_ldx_lift_batch(const char *path, int device, const uint64_t *pos, uint64_t n, uint64_t *out);

/* ---- the reference's on-disk r-index (<prefix>.thrbv.full.lcp.ms: moni_lcp::serialize / load, include/aligner/moni_lcp.hpp:178-225) ---- */
/* Layout of the absent r-index / sdsl pieces restated from recall (UNPINNED: the reference tree holds no such file); the reader takes a
 * file only if it parses to its last byte AND every redundancy in it agrees (moni_align_amd/csrc/ms_index_io.hpp).  Host-only calls. */
int moni_ms_file_info(const char *path, uint64_t *n, uint64_t *r);
/* Arrays for r runs (from moni_ms_file_info): F[256], heads[r], starts[r+1], ssa[r], esa[r], thr[r] (0 = none), slcp[r].  On
 * MONI_EIO err (if given) says which part of the file disagreed. */
int moni_ms_file_read(const char *path, uint64_t r, uint64_t *F, uint8_t *heads, uint64_t *starts, uint64_t *ssa, uint64_t *esa,
                      uint64_t *thr, uint64_t *slcp, char *err, uint64_t err_cap);
/* moni_lcp::serialize of a flat index (only n, r, F, heads, starts, ssa, esa, thr, slcp are read). */
int moni_ms_file_write(const moni_flat_index_t *idx, const char *path);
/* What aligner's constructor loads (seed_finder.hpp:66-124): <prefix>.thrbv.full.lcp.ms + <prefix>.ldx + the text.  text_path: plain
 * bytes (n - 1 of them, what PlainSlp::expandSubstr would return), or NULL - the output of `moni build` as it is: the .plain.slp grammar
 * (ShapedSlp, an absent submodule) is not read, the text is rebuilt from the BWT on the GPU instead (see moni_index_create). */
int moni_index_load_reference(const char *ms_path, const char *ldx_path, const char *text_path, int device, moni_index_t **out);

/* ---- measurement -------------------------------------------------------------------------- */
/* HIP-event time (ms) of the kernels of the last *_run on this ctx's stream.
 * which: 0 ms_lf, 1 mem_count, 2 mem_emit, 3 phi_count, 4 phi_emit, 5 extz, 6 whole run. */
int moni_last_kernel_ms(moni_ctx_t *ctx, int which, float *ms);
/* Work counters of the last moni_seed_run: out[0] LF steps, [1] threshold jumps, [2] phi steps,
 * [3] text bytes compared (the S, J, P, C of SURVEY.md §8(d)). */
int moni_last_counters(moni_ctx_t *ctx, uint64_t out[4]);
const char *moni_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MONI_HIP_H */
