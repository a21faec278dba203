// Host-side conversion of liftidx::lifts (the ones of levioSAM's ins / del bit-vectors, as moni_flat_index_t carries them) into
// the column-run form of lift_core.h.  Runs once at index load time.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/moni_hip.h"
#include "lift_core.h"

struct LiftTables {
    std::vector<moni_lift_seq_t> seqs;     // one per sequence
    std::vector<moni_lift_run_t> runs;
    std::vector<uint64_t> pdir;            // position directory (lift_core.h), (n_text >> MONI_PDIR_SHIFT) + 2 entries
    bool all_null = true;                  // every lift is a null lift onto its own sequence (FASTA-built index)

    // Returns MONI_OK or MONI_ERANGE / MONI_EINVAL with err set.
    int build(const moni_flat_index_t& f, std::string& err) {
        seqs.clear(); runs.clear(); all_null = true;
        const bool given = f.lift_second != nullptr;
        if (given && (!f.lift_len || !f.lift_ins_off || !f.lift_del_off)) { err = "incomplete lift arrays"; return MONI_EINVAL; }
        for (uint64_t i = 0; i < f.n_seq; ++i) {
            const uint64_t seq_len = f.seq_starts[i + 1] - f.seq_starts[i] - (f.seq_starts[i + 1] - f.seq_starts[i] >= f.w ? f.w : 0);
            moni_lift_seq_t S;
            S.second = given ? f.lift_second[i] : f.seq_starts[i];       // liftidx.hpp:150-157 null lift: lift(pos) = pos over text coordinates
            S.run_off = (uint32_t)runs.size();
            S.start = f.seq_starts[i]; S.end = f.seq_starts[i + 1];
            const uint64_t len = given ? f.lift_len[i] : seq_len;
            const uint64_t* ins = given && f.lift_ins ? f.lift_ins + f.lift_ins_off[i] : nullptr;
            const uint64_t* del = given && f.lift_del ? f.lift_del + f.lift_del_off[i] : nullptr;
            const uint64_t ni = given ? f.lift_ins_off[i + 1] - f.lift_ins_off[i] : 0, nd = given ? f.lift_del_off[i + 1] - f.lift_del_off[i] : 0;
            if (len >= (1ull << 32) - 1) { err = "lift with 2^32 or more alignment columns"; return MONI_ERANGE; }
            for (uint64_t k = 0; k < ni; ++k) if (ins[k] >= len || (k && ins[k] <= ins[k - 1])) { err = "lift ins positions not increasing inside the alignment"; return MONI_ERANGE; }
            for (uint64_t k = 0; k < nd; ++k) if (del[k] >= len || (k && del[k] <= del[k - 1])) { err = "lift del positions not increasing inside the alignment"; return MONI_ERANGE; }
            if (ni || nd || S.second != f.seq_starts[i]) all_null = false;
            // breakpoints: the ends of every maximal block of consecutive ones of either vector
            std::vector<uint64_t> bp;
            bp.push_back(0);
            auto blocks = [&](const uint64_t* v, uint64_t nv) {
                for (uint64_t k = 0; k < nv;) { uint64_t e = k + 1; while (e < nv && v[e] == v[e - 1] + 1) ++e; bp.push_back(v[k]); bp.push_back(v[e - 1] + 1); k = e; }
            };
            blocks(ins, ni); blocks(del, nd);
            bp.push_back(len);
            std::sort(bp.begin(), bp.end());
            bp.erase(std::unique(bp.begin(), bp.end()), bp.end());
            uint64_t hap = 0, ref = 0, pi = 0, pd = 0;
            uint32_t prev_flags = 0xFFFFFFFFu;
            for (size_t b = 0; b + 1 < bp.size(); ++b) {
                const uint64_t c0 = bp[b], c1 = bp[b + 1];
                while (pi < ni && ins[pi] < c0) ++pi;
                while (pd < nd && del[pd] < c0) ++pd;
                const uint32_t fl = ((pi < ni && ins[pi] == c0) ? MONI_LIFT_INS : 0u) | ((pd < nd && del[pd] == c0) ? MONI_LIFT_DEL : 0u);
                if (fl != prev_flags) { runs.push_back(moni_lift_run_t{(uint32_t)c0, (uint32_t)hap, (uint32_t)ref, fl}); prev_flags = fl; }
                if (!(fl & MONI_LIFT_DEL)) hap += c1 - c0;
                if (!(fl & MONI_LIFT_INS)) ref += c1 - c0;
            }
            if (runs.size() == S.run_off) runs.push_back(moni_lift_run_t{0, 0, 0, 0});          // no columns at all
            runs.push_back(moni_lift_run_t{(uint32_t)len, (uint32_t)hap, (uint32_t)ref, 0});    // sentinel: past the end, plain matches
            S.n_runs = (uint32_t)(runs.size() - S.run_off);
            seqs.push_back(S);
        }
        if (runs.size() >= (1ull << 32)) { err = "too many lift runs"; return MONI_ERANGE; }
        // position directory
        const uint64_t n_text = f.n ? f.n - 1 : f.seq_starts[f.n_seq];
        const uint64_t nb = (n_text >> MONI_PDIR_SHIFT) + 2;
        pdir.assign(nb, 0);
        uint64_t sid = 0; uint32_t k = seqs.empty() ? 0 : seqs[0].run_off;
        for (uint64_t b = 0; b < nb; ++b) {
            const uint64_t pos = b << MONI_PDIR_SHIFT;
            bool moved = false;
            while (sid + 1 < f.n_seq && pos >= f.seq_starts[sid + 1]) { ++sid; moved = true; }
            if (moved) k = seqs[sid].run_off;
            const uint64_t st = pos >= f.seq_starts[sid] ? pos - f.seq_starts[sid] : 0;
            const uint32_t last = seqs[sid].run_off + seqs[sid].n_runs;
            while (k + 1 < last && (uint64_t)runs[k + 1].hap <= st) ++k;
            pdir[b] = (uint64_t)sid | ((uint64_t)k << 32);
        }
        return MONI_OK;
    }
};
