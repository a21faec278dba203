// moni-hip-align: the `align_full_ksw2` command line (src/align/align_full_ksw2.cpp:101-431) over libmoni_hip.so.
//
//   moni-hip-align <prefix> -p reads.fq [-o out.sam] [-t T] [-b B] [-l len] [-L ext] [-A a] [-B b] [-O o1[,o2]] [-E e1[,e2]]
//                  [-s] [-f] [-a] [-S n] [-F f] [-w iter] [-v pred] [-x dx] [-y dy] [-k mem] [-j score] [--gpus N] [--gpu-batch R]
//
// Same getopt string, flag meanings, struct defaults and default output name as the reference, so the `moni align` wrapper
// (pipeline/moni.in:494-546) can call it unchanged.  Index: <prefix>.mfi (the flat arrays moni_align_amd/index_build.py
// writes; reading the reference's sdsl-serialised files is SURVEY §8(f) item 1).  Reads are streamed in large batches (the
// reference's -b is a per-thread batch of 512; a GPU wants ~10^5), one worker thread and one index replica per GPU, output
// written in input order.  Not implemented here (exit 1 with a message): paired-end (-1/-2), -m, -c, -q, -n, -Z.
#include <getopt.h>
#include <libgen.h>
#include <zlib.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/moni_hip.h"

static void die(const std::string& msg) { fprintf(stderr, "[ERROR] %s\n", msg.c_str()); exit(1); }   // common.hpp:108-117
static void info(const std::string& msg) { printf("[INFO] Message: %s\n", msg.c_str()); fflush(stdout); }

struct Batch {
    std::vector<uint8_t> seq, qual, names;
    std::vector<uint64_t> off{0}, name_off{0};
    bool has_qual = true;
    size_t n() const { return off.size() - 1; }
};

// kseq.h semantics: name = first word of the header, sequence/quality may span lines, '>' or '@' records
struct Reader {
    gzFile fp;
    std::string line;
    bool have_line = false, eof = false;
    explicit Reader(const std::string& path) { fp = gzopen(path.c_str(), "r"); if (!fp) die("open() file " + path + " failed"); }
    ~Reader() { if (fp) gzclose(fp); }
    bool getline() {
        if (have_line) { have_line = false; return true; }
        line.clear();
        char buf[65536];
        bool any = false;
        while (gzgets(fp, buf, sizeof buf)) {
            any = true;
            size_t l = strlen(buf);
            if (l && buf[l - 1] == '\n') { buf[--l] = 0; if (l && buf[l - 1] == '\r') buf[--l] = 0; line.append(buf, l); return true; }
            line.append(buf, l);
        }
        if (!any) eof = true;
        return any;
    }
    // appends one record; false at end of file
    bool next(Batch& b) {
        while (getline()) if (!line.empty() && (line[0] == '>' || line[0] == '@')) break;
        if (eof && line.empty()) return false;
        if (line.empty() || (line[0] != '>' && line[0] != '@')) return false;
        const bool fq = line[0] == '@';
        size_t e = 1;
        while (e < line.size() && !isspace((unsigned char)line[e])) ++e;
        b.names.insert(b.names.end(), line.begin() + 1, line.begin() + e);
        b.name_off.push_back(b.names.size());
        size_t slen = 0;
        while (getline()) {
            if (!line.empty() && (line[0] == '+' || line[0] == '>' || (line[0] == '@' && !fq))) break;
            if (!line.empty() && line[0] == '@' && fq) break;
            for (char c : line) if (!isspace((unsigned char)c)) { b.seq.push_back((uint8_t)c); ++slen; }
        }
        b.off.push_back(b.seq.size());
        if (!eof && !line.empty() && line[0] == '+') {
            size_t qlen = 0;
            while (qlen < slen && getline()) { for (char c : line) { b.qual.push_back((uint8_t)c); ++qlen; } }
            if (qlen != slen) die("truncated quality string");
        } else {
            if (!eof) have_line = true;            // header of the next record
            b.has_qual = false;
            b.qual.resize(b.seq.size(), (uint8_t)'*');
        }
        return true;
    }
};

struct Args {
    std::string filename, patterns, mate1, mate2, output;
    size_t b = 512, th = 1;
    moni_align_params_t P;
    bool report_mems = false, csv = false, no_lcp = false, shaped_slp = false, secondary = false;
    int gpus = 1;
    size_t gpu_batch = 262144;
    bool dry_run = false;
};

static void parse(int argc, char** argv, Args& a) {
    moni_align_params_default(&a.P);
    a.P.n_seeds_thr = 5000; a.P.freq_thr = 0.30;          // struct defaults of align_full_ksw2.cpp:70,73 (the wrapper always passes -S/-F)
    std::vector<char*> av;
    for (int i = 0; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { a.gpus = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "--gpu-batch") && i + 1 < argc) { a.gpu_batch = strtoull(argv[++i], nullptr, 10); continue; }
        if (!strcmp(argv[i], "--dry-run")) { a.dry_run = true; continue; }
        av.push_back(argv[i]);
    }
    const std::string usage = "usage: " + std::string(argv[0]) + " infile [-p patterns] [-o output] [-t threads] [-b batch] [-l len] [-L ext_l] [-A smatch] "
                              "[-B smismatch] [-O gapo] [-E gape] [-s seeds_dis] [-f freq_dis] [-S seeds_thr] [-F freq_thr] [-w max_iter] [-v max_pred] "
                              "[-x max_dist_x] [-y max_dist_y] [-k min_chain_mem] [-j min_chain_score] [-a chain_dis] [--gpus N] [--gpu-batch reads]\n";
    int c;
    char* s;
    optind = 1;
    while ((c = getopt((int)av.size(), av.data(), "ql:hp:o:t:1:2:b:A:B:O:E:L:dsfnD:S:F:w:v:x:y:k:j:Zaumc")) != -1) {
        switch (c) {
            case 'p': a.patterns = optarg; break;
            case 'o': a.output = optarg; break;
            case '1': a.mate1 = optarg; break;
            case '2': a.mate2 = optarg; break;
            case 'l': a.P.min_len = (uint32_t)std::stoi(optarg); break;
            case 'b': a.b = (size_t)std::stoi(optarg); break;
            case 't': a.th = (size_t)std::stoi(optarg); break;
            case 'L': a.P.ext_len = (uint32_t)std::stoi(optarg); break;
            case 'A': a.P.smatch = (int8_t)std::stoi(optarg); break;
            case 'B': a.P.smismatch = (int8_t)std::stoi(optarg); break;
            case 'd': break;                                   // filter_dir: paired-end only
            case 's': a.P.filter_seeds = 0; break;
            case 'f': a.P.filter_freq = 0; break;
            case 'n': a.no_lcp = true; break;
            case 'D': (void)std::stoi(optarg); break;          // dir_thr: paired-end only (parsed with stoi, align_full_ksw2.cpp:195-198)
            case 'S': a.P.n_seeds_thr = (uint32_t)std::stoi(optarg); break;
            case 'F': a.P.freq_thr = std::stod(optarg); break;
            case 'O': a.P.gapo = a.P.gapo2 = (int8_t)strtol(optarg, &s, 10); if (*s == ',') a.P.gapo2 = (int8_t)strtol(s + 1, &s, 10); break;
            case 'E': a.P.gape = a.P.gape2 = (int8_t)strtol(optarg, &s, 10); if (*s == ',') a.P.gape2 = (int8_t)strtol(s + 1, &s, 10); break;
            case 'q': a.shaped_slp = true; break;
            case 'w': a.P.max_iter = std::stoi(optarg); break;
            case 'v': a.P.max_pred = std::stoi(optarg); break;
            case 'x': a.P.max_dist_x = std::stoi(optarg); break;
            case 'y': a.P.max_dist_y = std::stoi(optarg); break;
            case 'k': a.P.min_chain_length = std::stoi(optarg); break;
            case 'j': a.P.min_chain_score = std::stoi(optarg); break;
            case 'Z': a.secondary = true; break;
            case 'a': a.P.left_mem_check = 0; break;
            case 'u': break;                                   // find_orphan: paired-end only
            case 'm': a.report_mems = true; break;
            case 'c': a.csv = true; break;
            case 'h': die(usage);
            default: die("Unknown option.\n" + usage);
        }
    }
    if ((int)av.size() == optind + 1) a.filename = av[optind];
    else die("Invalid number of arguments\n" + usage);
    if (a.th > 1) a.P.host_threads = (uint32_t)a.th;
}

int main(int argc, char** argv) {
    Args a;
    parse(argc, argv, a);
    if (!a.mate1.empty() || !a.mate2.empty()) die("paired-end alignment (-1/-2) is not implemented in moni-hip-align yet");
    if (a.report_mems || a.csv || a.no_lcp || a.shaped_slp || a.secondary) die("options -m, -c, -n, -q, -Z are not implemented in moni-hip-align yet");
    if (a.patterns.empty()) die("no reads given (-p)");
    std::string fn = a.filename;
    std::vector<char> tmp(fn.begin(), fn.end()); tmp.push_back(0);
    const std::string base_name = basename(tmp.data());
    std::string sam_filename = a.patterns + "_" + base_name + "_" + std::to_string(a.P.min_len) + ".sam";   // align_full_ksw2.cpp:347-349
    if (!a.output.empty()) sam_filename = a.output;
    info("Output file: " + sam_filename);
    if (a.dry_run) {
        Reader rd(a.patterns);
        Batch b; size_t n = 0, bases = 0;
        while (rd.next(b)) { ++n; }
        bases = b.seq.size();
        printf("dry-run: reads=%zu bases=%zu min_len=%u ext_len=%u S=%u F=%.2f O=%d,%d E=%d,%d threads=%zu gpus=%d out=%s first=%.*s\n", n, bases, a.P.min_len,
               a.P.ext_len, a.P.n_seeds_thr, a.P.freq_thr, a.P.gapo, a.P.gapo2, a.P.gape, a.P.gape2, a.th, a.gpus, sam_filename.c_str(),
               n ? (int)b.name_off[1] : 0, n ? (const char*)b.names.data() : "");
        return 0;
    }
    const std::string idx_path = a.filename + ".mfi";
    std::vector<moni_index_t*> idx(a.gpus, nullptr);
    std::vector<moni_ctx_t*> ctx(a.gpus, nullptr);
    for (int g = 0; g < a.gpus; ++g) {
        if (moni_index_load(idx_path.c_str(), g, &idx[g])) die("cannot load " + idx_path + " on GPU " + std::to_string(g) + " (moni-hip has no CPU path)");
        if (moni_ctx_create(idx[g], &ctx[g])) die("cannot create a context on GPU " + std::to_string(g));
    }
    FILE* out = fopen(sam_filename.c_str(), "w");
    if (!out) die("open() file " + sam_filename + " failed");
    { char* h; uint64_t hl; if (moni_sam_header(idx[0], &h, &hl)) die("header"); fwrite(h, 1, hl, out); moni_free(h); }
    auto t0 = std::chrono::steady_clock::now();
    Reader rd(a.patterns);
    std::mutex mu_in, mu_out;
    std::condition_variable cv_out;
    size_t next_in = 0, next_out = 0, processed = 0, aligned = 0;
    std::map<size_t, std::string> done;
    auto worker = [&](int g) {
        while (true) {
            Batch b;
            size_t id;
            {
                std::lock_guard<std::mutex> lk(mu_in);                    // mt_kbseq_read (align_reads_dispatcher.hpp:74-84)
                while (b.n() < a.gpu_batch && rd.next(b)) {}
                if (b.n() == 0) return;
                id = next_in++;
            }
            moni_read_batch_t rb{b.seq.data(), b.off.data(), b.n()};
            char* sam; uint64_t len; moni_align_stats_t st;
            if (moni_align_batch(ctx[g], &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &sam, &len, &st))
                die("moni_align_batch failed");
            std::unique_lock<std::mutex> lk(mu_out);
            done[id] = std::string(sam, len);
            moni_free(sam);
            processed += st.reads; aligned += st.aligned;
            while (!done.empty() && done.begin()->first == next_out) { fwrite(done.begin()->second.data(), 1, done.begin()->second.size(), out); done.erase(done.begin()); ++next_out; }
        }
    };
    std::vector<std::thread> th;
    for (int g = 0; g < a.gpus; ++g) th.emplace_back(worker, g);
    for (auto& t : th) t.join();
    fclose(out);
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info("Number of aligned reads: " + std::to_string(aligned) + "/" + std::to_string(processed));
    info("Elapsed time (s): " + std::to_string(el));
    for (int g = 0; g < a.gpus; ++g) { moni_ctx_destroy(ctx[g]); moni_index_destroy(idx[g]); }
    return 0;
}
