// pe_core.h compiled for the host with large capacities, in its own namespace (the structs and inline functions of align_core.h / pe_core.h
// exist in moni_hip.hip's translation unit with the kernels' capacities).  See pe_big.h.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <memory>

#include "../../include/moni_hip.h"
#include "sort_emul.h"
#include "lift_core.h"
#include "pe_big.h"

#define AC_MAX_MEMS 4096
#define AC_MAX_ANCH 32768
#define AC_MAX_CHAINS 8192
#define AC_MAX_POOL 65536
#define AC_MAX_BEST 1024
#define AC_MAX_LEFT 8192
#define AC_MAX_ALT 2048
#define AC_MAX_FILL 128
#define AC_MAX_CIGAR 8192
#define PE_MAX_BEST 1024

#define PE_CSV_COUNT
namespace pe_big {
#include "align_core.h"
#include "pe_core.h"
}  // namespace pe_big

int pe_big_run(const void* pe_params, size_t pe_params_size, const moni_mem_t* mems, const uint64_t* rmo, const uint32_t* aux, const uint64_t* occs,
               const uint64_t* offs, std::vector<PeBigPair>& pairs, const PeBigDp& dp) {
    using namespace pe_big;
    if (pe_params_size != sizeof(pe_params_t)) return MONI_EINVAL;          // the two translation units must agree on the layout
    pe_params_t PP;
    memcpy(&PP, pe_params, sizeof PP);
    const size_t GROUP = 64;                                                // states alive at once (a few MB each)
    std::vector<moni_dp_task_t> tasks;
    std::vector<moni_dp_result_t> res;
    std::vector<uint32_t> cig;
    for (size_t g0 = 0; g0 < pairs.size(); g0 += GROUP) {
        const size_t g1 = std::min(pairs.size(), g0 + GROUP);
        std::vector<std::unique_ptr<pe_ws_t>> st(g1 - g0);
        std::vector<size_t> first(g1 - g0 + 1, 0);
        for (size_t i = g0; i < g1; ++i) {
            st[i - g0].reset(new pe_ws_t);
            pe_ws_t& W = *st[i - g0];
            const uint64_t p = pairs[i].pair;
            bool too_long = false;
            for (int k = 0; k < 2; ++k) {
                W.off[k] = offs[2 * p + k]; W.m[k] = (uint32_t)(offs[2 * p + k + 1] - offs[2 * p + k]);
                W.min_score_m[k] = W.m[k] ? (int32_t)(20 + 8 * log((double)W.m[k])) : INT32_MIN;
                too_long = too_long || W.m[k] >= 32768;                    // 16-bit read coordinates in the chaining nodes
            }
            W.min_score = (int32_t)((uint32_t)W.min_score_m[0] + (uint32_t)W.min_score_m[1]);
            if (too_long || W.m[0] == 0 || W.m[1] == 0) { W.csv_filter = W.csv_skipped = 0; ac_reset(W.W); W.W.overflow = too_long ? 1u : 0u; W.final = pe_pscore_t(); W.score2 = W.score2_m[0] = W.score2_m[1] = 0; W.sub_n = 0; W.strand = 0; W.filled[0] = W.filled[1] = 0; W.n_alt[0] = W.n_alt[1] = 0; continue; }
            if (pe_init(W, PP, mems, rmo, aux, occs, p)) pe_drive(W, PP, nullptr, nullptr);
        }
        while (true) {                                                      // DP rounds of the group
            tasks.clear();
            for (size_t i = 0; i < st.size(); ++i) {
                pe_ws_t& W = *st[i];
                first[i] = tasks.size();
                if (!W.W.overflow && W.W.stage != AC_DONE) tasks.insert(tasks.end(), W.W.tasks, W.W.tasks + W.W.n_tasks);
            }
            first[st.size()] = tasks.size();
            if (tasks.empty()) break;
            const int rc = dp(tasks, res, cig);
            if (rc) return rc;
            for (size_t i = 0; i < st.size(); ++i) {
                pe_ws_t& W = *st[i];
                if (W.W.overflow || W.W.stage == AC_DONE || first[i + 1] == first[i]) continue;
                pe_drive(W, PP, res.data() + first[i], cig.data());        // results index the pair's own tasks; cigar_off indexes the round's pool
            }
        }
        for (size_t i = g0; i < g1; ++i) {
            const pe_ws_t& W = *st[i - g0];
            PeBigPair& R = pairs[i];
            R.status = W.W.overflow ? 2u : (W.W.aligned ? 1u : 0u);
            R.strand = W.strand; R.tot = W.final.tot; R.score2 = W.score2; R.sub_n = W.sub_n; R.dist = W.final.dist;
            R.mate_score[0] = W.final.m1.score; R.mate_score[1] = W.final.m2.score;
            R.csv_filter = W.csv_filter; R.csv_skipped = W.csv_skipped;
            for (int k = 0; k < 2; ++k) {
                R.score2_m[k] = W.score2_m[k];
                R.filled[k] = 0; R.orphan[k] = 0; R.cig[k].clear(); R.alt_pos[k].clear(); R.alt_score[k].clear();
                if (R.status == 1 && PP.finalize && W.filled[k]) {
                    R.filled[k] = 1; R.orphan[k] = W.orphan[k]; R.ref_pos[k] = W.ref_pos[k]; R.as[k] = W.as[k];
                    R.cig[k].assign(W.cigar[k], W.cigar[k] + W.n_cigar[k]);
                    R.alt_pos[k].assign(W.alt_pos[k], W.alt_pos[k] + W.n_alt[k]);
                    R.alt_score[k].assign(W.alt_score[k], W.alt_score[k] + W.n_alt[k]);
                }
            }
        }
    }
    return MONI_OK;
}
