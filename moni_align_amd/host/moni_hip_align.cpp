// moni-hip-align: the `align_full_ksw2` command line (src/align/align_full_ksw2.cpp:101-431) over libmoni_hip.so.
//
//   moni-hip-align <prefix> -p reads.fq [-o out.sam] [-t T] [-b B] [-l len] [-L ext] [-A a] [-B b] [-O o1[,o2]] [-E e1[,e2]]
//                  [-s] [-f] [-a] [-S n] [-F f] [-w iter] [-v pred] [-x dx] [-y dy] [-k mem] [-j score] [--gpus N] [--gpu-batch R]
//
// Same getopt string, flag meanings, struct defaults and default output name as the reference, so the `moni align` wrapper
// (pipeline/moni.in:494-546) can call it unchanged.  Index: <prefix>.mfi (the flat arrays moni_align_amd/index_build.py
// writes) or, when that file is absent, the reference's own files <prefix>.thrbv.full.lcp.ms + <prefix>.ldx as `moni build` leaves them
// (moni_index_load_reference: the .plain.slp grammar is not read, the text is rebuilt from the BWT on the GPU; a <prefix>.txt with the
// text as plain bytes is used when present).  The .thrbv.full.lcp.ms reader follows the sdsl / r-index layouts from recall: no file
// written by the reference was available to check it against.  Reads are streamed in large batches (the
// reference's -b is a per-thread batch of 512; a GPU wants ~10^6).  Plain FASTQ / FASTA files (the streaming path, fast_align below):
// the file is mapped and cut into record-aligned byte ranges of about --gpu-batch reads; three workers per GPU, each with its own
// context, take ranges in order: parse (two passes over the range with a few helper threads: count, then fill the batch arrays in
// place), moni_align_stream (upload, kernels, the lines in read order in the context's pinned buffer), then pwrite of the block at its
// place in the output file, again in parallel slices.  A block's place is known as soon as the earlier ranges have reported their
// lengths, so parsing, GPU work and file writes of different ranges overlap and nothing is copied twice.  gzip input, -m and the legacy
// modes take the older path: a reader thread parses ahead into a bounded queue, two workers per GPU, a writer thread with a bounded
// in-order window.  -m writes the report-MEMs records (aligner_ksw2.hpp:346-373);
// --ms / --mems write the legacy `moni ms` / `moni mems` text outputs (src/matching_statistics.cpp:520-610, src/mems.cpp:520-600).
// -n loads <prefix>.thrbv.full.ms (no LCP samples); -q is accepted (the text comes from the BWT, not from either grammar).
// -c writes <sam>.csv (per-read MEM statistics, csv.hpp:55-67; through the host pipeline - for pairs one line per pair, moni_pe_align_csv_batch).  -Z (secondary chains,
// chain.hpp:442-727) acts on paired input and is ignored for single-end input, as in the reference (aligner_ksw2.hpp:1190-1191 is the only call site).
#include <fcntl.h>
#include <getopt.h>
#include <libgen.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cerrno>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/moni_hip.h"

static void die(const std::string& msg) { fprintf(stderr, "[ERROR] %s\n", msg.c_str()); exit(1); }   // common.hpp:108-117
// output that did not reach the file (a full disk) is an error, not a shorter SAM file with exit code 0
static void put(const void* p, size_t n, FILE* f) { if (n && fwrite(p, 1, n, f) != n) die("short write to the output file"); }
static void close_out(FILE* f) { if (f && fclose(f) != 0) die("closing the output file failed (output incomplete)"); }
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

// Plain (uncompressed) four-line FASTQ / two-line FASTA mapped into memory: records are split with memchr.  Anything else (gzip, wrapped
// sequences) takes the kseq-style Reader above.
struct MappedReader {
    const char* p = nullptr; size_t n = 0, at = 0; int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 4) { ::close(fd); fd = -1; return false; }
        n = (size_t)st.st_size;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        madvise(m, n, MADV_SEQUENTIAL);
        p = (const char*)m;
        if ((unsigned char)p[0] == 0x1f && (unsigned char)p[1] == 0x8b) { close(); return false; }     // gzip
        // the fast path needs one-line sequences: check the first record
        const char* l1 = (const char*)memchr(p, '\n', n);
        if (!l1 || (p[0] != '@' && p[0] != '>')) { close(); return false; }
        const char* l2 = (const char*)memchr(l1 + 1, '\n', n - (size_t)(l1 + 1 - p));
        if (!l2) { close(); return false; }
        if (p[0] == '@' && (l2 + 1 >= p + n || l2[1] != '+')) { close(); return false; }
        return true;
    }
    void close() { if (p) munmap((void*)p, n); p = nullptr; if (fd >= 0) ::close(fd); fd = -1; }
    ~MappedReader() { close(); }
    static const char* eol(const char* s, const char* end) { const char* e = (const char*)memchr(s, '\n', (size_t)(end - s)); return e ? e : end; }
    bool next(Batch& b) {
        const char* end = p + n;
        const char* s = p + at;
        while (s < end && (*s == '\n' || *s == '\r')) ++s;
        if (s >= end) return false;
        const bool fq = *s == '@';
        if (!fq && *s != '>') die("malformed record in the reads file");
        const char* e1 = eol(s, end);
        const char* w = s + 1;
        while (w < e1 && !isspace((unsigned char)*w)) ++w;
        b.names.insert(b.names.end(), (const uint8_t*)s + 1, (const uint8_t*)w);
        b.name_off.push_back(b.names.size());
        const char* q = e1 < end ? e1 + 1 : end;
        const char* e2 = eol(q, end);
        size_t slen = (size_t)(e2 - q);
        if (slen && q[slen - 1] == '\r') --slen;
        b.seq.insert(b.seq.end(), (const uint8_t*)q, (const uint8_t*)q + slen);
        b.off.push_back(b.seq.size());
        const char* nx = e2 < end ? e2 + 1 : end;
        if (fq) {
            const char* e3 = eol(nx, end);                     // '+' line
            const char* ql = e3 < end ? e3 + 1 : end;
            const char* e4 = eol(ql, end);
            size_t qlen = (size_t)(e4 - ql);
            if (qlen && ql[qlen - 1] == '\r') --qlen;
            if (qlen != slen) die("truncated quality string");
            b.qual.insert(b.qual.end(), (const uint8_t*)ql, (const uint8_t*)ql + qlen);
            nx = e4 < end ? e4 + 1 : end;
        } else {
            b.has_qual = false;
            b.qual.resize(b.seq.size(), (uint8_t)'*');
        }
        at = (size_t)(nx - p);
        return true;
    }
};

struct Args {
    std::string filename, patterns, mate1, mate2, output;
    size_t b = 512, th = 1;
    moni_align_params_t P;
    bool report_mems = false, csv = false, no_lcp = false, shaped_slp = false, secondary = false;
    bool legacy_ms = false, legacy_mems = false;      // --ms / --mems
    int gpus = 1;
    size_t gpu_batch = 1048576;
    int ctx_per_gpu = 3;               // streaming path: contexts (ranges in flight) per GPU
    bool dry_run = false;
    bool dry_write = false;     // --dry-run-write: no GPU, but the single-end front end's batching runs - ranges, workers of all (absent) GPUs, blocks placed in input order - with placeholder records
    moni_pe_params_t PE;               // -d, -D (paired-end)
    bool find_orphan = true;           // -u switches orphan recovery off
};

static void parse(int argc, char** argv, Args& a) {
    moni_align_params_default(&a.P);
    moni_pe_params_default(&a.PE);
    a.P.n_seeds_thr = 5000; a.P.freq_thr = 0.30;          // struct defaults of align_full_ksw2.cpp:70,73 (the wrapper always passes -S/-F)
    std::vector<char*> av;
    for (int i = 0; i < argc; ++i) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) { a.gpus = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "--gpu-batch") && i + 1 < argc) { a.gpu_batch = strtoull(argv[++i], nullptr, 10); continue; }
        if (!strcmp(argv[i], "--dry-run")) { a.dry_run = true; continue; }
        if (!strcmp(argv[i], "--dry-run-write")) { a.dry_run = true; a.dry_write = true; continue; }
        if (!strcmp(argv[i], "--ctx-per-gpu") && i + 1 < argc) { a.ctx_per_gpu = std::max(1, atoi(argv[++i])); continue; }
        if (!strcmp(argv[i], "--ms")) { a.legacy_ms = true; continue; }
        if (!strcmp(argv[i], "--mems")) { a.legacy_mems = true; continue; }
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
            case 'd': a.PE.filter_dir = 0; break;              // filter_dir off (align_full_ksw2.cpp:189-191)
            case 's': a.P.filter_seeds = 0; break;
            case 'f': a.P.filter_freq = 0; break;
            case 'n': a.no_lcp = true; break;
            case 'D': a.PE.dir_thr = std::stoi(optarg); break;  // dir_thr (parsed with stoi, align_full_ksw2.cpp:195-198)
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
            case 'u': a.find_orphan = false; break;
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


// ---- streaming path for plain files ---------------------------------------------------------------------------------------------------
namespace fastpath {

static inline const char* next_line(const char* s, const char* end) { const char* e = (const char*)memchr(s, '\n', (size_t)(end - s)); return e ? e + 1 : end; }

// the first record that starts at or after s.  Four-line FASTQ: a line that starts with '@' whose second-next line starts with '+'
// (a quality line may start with '@', but then the second-next line is a sequence); FASTA: a line that starts with '>'.
static const char* align_record(const char* s, const char* base, const char* end, bool fq) {
    if (s <= base) return base;
    if (s >= end) return end;
    s = next_line(s - 1, end);
    while (s < end) {
        if (fq) {
            if (*s == '@') { const char* l3 = next_line(next_line(s, end), end); if (l3 < end && *l3 == '+') return s; }
        } else if (*s == '>') return s;
        s = next_line(s, end);
    }
    return end;
}

struct Count { size_t n = 0, seq = 0, name = 0; };
struct Raw {          // uninitialised grow-only bytes (a vector would zero 300 MB per batch)
    uint8_t* p = nullptr; size_t cap = 0;
    void ensure(size_t n) { if (n > cap) { free(p); cap = n + n / 8 + 64; p = (uint8_t*)malloc(cap); if (!p) die("out of memory"); } }
    ~Raw() { free(p); }
};

// one pass over the records of [a, b): FILL = false counts, FILL = true writes them at the given cursors
template <bool FILL>
static Count walk(const char* a, const char* b, bool fq, uint8_t* seq, uint8_t* qual, uint8_t* names, uint64_t* off, uint64_t* name_off,
                  uint64_t seq_at, uint64_t name_at) {
    Count c;
    const char* s = a;
    while (s < b) {
        while (s < b && (*s == '\n' || *s == '\r')) ++s;
        if (s >= b) break;
        if (*s != (fq ? '@' : '>')) die("malformed record in the reads file");
        const char* e1 = (const char*)memchr(s, '\n', (size_t)(b - s)); if (!e1) e1 = b;
        const char* w = s + 1;
        while (w < e1 && !isspace((unsigned char)*w)) ++w;
        const size_t nl = (size_t)(w - (s + 1));
        const char* q = e1 < b ? e1 + 1 : b;
        const char* e2 = (const char*)memchr(q, '\n', (size_t)(b - q)); if (!e2) e2 = b;
        size_t sl = (size_t)(e2 - q);
        if (sl && q[sl - 1] == '\r') --sl;
        const char* nx = e2 < b ? e2 + 1 : b;
        const char* ql = nullptr;
        if (fq) {
            const char* e3 = (const char*)memchr(nx, '\n', (size_t)(b - nx)); if (!e3) e3 = b;
            ql = e3 < b ? e3 + 1 : b;
            const char* e4 = (const char*)memchr(ql, '\n', (size_t)(b - ql)); if (!e4) e4 = b;
            size_t qn = (size_t)(e4 - ql);
            if (qn && ql[qn - 1] == '\r') --qn;
            if (qn != sl) die("truncated quality string");
            nx = e4 < b ? e4 + 1 : b;
        }
        if (FILL) {
            memcpy(names + name_at + c.name, s + 1, nl);
            memcpy(seq + seq_at + c.seq, q, sl);
            if (fq) memcpy(qual + seq_at + c.seq, ql, sl); else memset(qual + seq_at + c.seq, '*', sl);
            off[c.n + 1] = seq_at + c.seq + sl; name_off[c.n + 1] = name_at + c.name + nl;
        }
        c.n++; c.seq += sl; c.name += nl;
        s = nx;
    }
    return c;
}

struct ParsedBatch {
    Raw seq, qual, names;
    std::vector<uint64_t> off, name_off;
    size_t n = 0;
};

static void parse_range(const char* a, const char* b, const char* base, const char* end, bool fq, int P, ParsedBatch& B) {
    std::vector<const char*> cut(P + 1);
    cut[0] = a; cut[P] = b;
    for (int i = 1; i < P; ++i) { cut[i] = align_record(a + (size_t)(b - a) / P * i, base, end, fq); if (cut[i] > b) cut[i] = b; if (cut[i] < cut[i - 1]) cut[i] = cut[i - 1]; }
    std::vector<Count> cnt(P);
    auto run = [&](auto&& fn) {
        std::vector<std::thread> th;
        for (int i = 1; i < P; ++i) th.emplace_back(fn, i);
        fn(0);
        for (auto& t : th) t.join();
    };
    run([&](int i) { cnt[i] = walk<false>(cut[i], cut[i + 1], fq, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0); });
    std::vector<size_t> n_at(P + 1, 0), s_at(P + 1, 0), m_at(P + 1, 0);
    for (int i = 0; i < P; ++i) { n_at[i + 1] = n_at[i] + cnt[i].n; s_at[i + 1] = s_at[i] + cnt[i].seq; m_at[i + 1] = m_at[i] + cnt[i].name; }
    B.n = n_at[P];
    B.seq.ensure(s_at[P] + 16); B.qual.ensure(s_at[P] + 16); B.names.ensure(m_at[P] + 16);
    if (B.off.size() < B.n + 1) { B.off.resize(B.n + 1 + B.n / 8); B.name_off.resize(B.n + 1 + B.n / 8); }
    B.off[0] = 0; B.name_off[0] = 0;
    run([&](int i) { walk<true>(cut[i], cut[i + 1], fq, B.seq.p, B.qual.p, B.names.p, B.off.data() + n_at[i], B.name_off.data() + n_at[i], s_at[i], m_at[i]); });
}

static void pwrite_all(int fd, const char* p, size_t n, uint64_t at) {
    while (n) {
        const ssize_t w = pwrite(fd, p, n, (off_t)at);
        if (w < 0) { if (errno == EINTR) continue; die("write to the output file failed"); }
        p += w; n -= (size_t)w; at += (uint64_t)w;
    }
}

// the end of the n_rec-th record from s in a mapped four-line FASTQ file (or the end of the file; got = the records before it): the newlines of a
// window are counted by P threads side by side, the slice in which the count is reached is scanned
static const char* skip_records(const char* s, const char* end, size_t n_rec, int P, size_t& got) {
    size_t need = 4 * n_rec, lines = 0;
    const char* cur = s;
    while (need > 0 && cur < end) {
        const size_t win = std::min<size_t>((size_t)(end - cur), std::max<size_t>((size_t)1 << 20, need * 48));
        std::vector<size_t> cnt(P, 0);
        auto count = [&](int i) {
            const char* a = cur + win / P * i; const char* b = i + 1 == P ? cur + win : cur + win / P * (i + 1);
            size_t c = 0;
            while (a < b) { const char* e = (const char*)memchr(a, '\n', (size_t)(b - a)); if (!e) break; ++c; a = e + 1; }
            cnt[i] = c;
        };
        { std::vector<std::thread> th; for (int i = 1; i < P; ++i) th.emplace_back(count, i); count(0); for (auto& t : th) t.join(); }
        size_t total = 0;
        for (int i = 0; i < P; ++i) total += cnt[i];
        if (total < need) { need -= total; lines += total; cur += win; continue; }
        int i = 0;
        while (cnt[i] < need) { need -= cnt[i]; lines += cnt[i]; ++i; }
        const char* a = cur + win / P * i;
        while (need > 0) { const char* e = (const char*)memchr(a, '\n', (size_t)(end - a)); a = e + 1; --need; ++lines; }
        cur = a;
    }
    if (cur >= end && end > s && end[-1] != '\n') ++lines;          // a last line without its newline
    got = lines / 4;
    return cur;
}

// n_pairs pairs from the cursors of two mapped four-line FASTQ files: both ranges parsed in two passes by P threads each (parse_range), then
// interleaved (mates 2p, 2p + 1) into uninitialised buffers
struct PairBatch { Raw seq, qual, names; std::vector<uint64_t> off, name_off; size_t n_reads = 0; };
static size_t read_pairs_fast(const char* base[2], const char* end[2], size_t at[2], size_t n_pairs, int P, ParsedBatch pb[2], PairBatch& B) {
    size_t got[2] = {0, 0};
    const char* lim[2];
    auto one = [&](int k) {
        lim[k] = skip_records(base[k] + at[k], end[k], n_pairs, P, got[k]);
        if (got[k]) parse_range(base[k] + at[k], lim[k], base[k], end[k], true, P, pb[k]); else pb[k].n = 0;
    };
    { std::thread t(one, 1); one(0); t.join(); }
    if (got[0] != got[1]) die("the mate files have different numbers of records");
    if (pb[0].n != got[0] || pb[1].n != got[1]) die("the mate files are not four-line FASTQ throughout (blank or wrapped lines): compress them or use FASTA to take the record-by-record reader");
    at[0] = (size_t)(lim[0] - base[0]); at[1] = (size_t)(lim[1] - base[1]);
    const size_t n = got[0];
    if (!n) return 0;
    B.n_reads = 2 * n;
    if (B.off.size() < 2 * n + 1) { B.off.resize(2 * n + 1); B.name_off.resize(2 * n + 1); }
    const ParsedBatch& b1 = pb[0]; const ParsedBatch& b2 = pb[1];
    B.seq.ensure(b1.off[n] + b2.off[n] + 16); B.qual.ensure(b1.off[n] + b2.off[n] + 16); B.names.ensure(b1.name_off[n] + b2.name_off[n] + 16);
    auto part = [&](size_t lo, size_t hi) {
        for (size_t p = lo; p < hi; ++p) {
            const uint64_t o1 = b1.off[p] + b2.off[p], o2 = b1.off[p + 1] + b2.off[p], m1 = b1.name_off[p] + b2.name_off[p], m2 = b1.name_off[p + 1] + b2.name_off[p];
            B.off[2 * p] = o1; B.off[2 * p + 1] = o2; B.name_off[2 * p] = m1; B.name_off[2 * p + 1] = m2;
            const size_t l1 = (size_t)(b1.off[p + 1] - b1.off[p]), l2 = (size_t)(b2.off[p + 1] - b2.off[p]);
            memcpy(B.seq.p + o1, b1.seq.p + b1.off[p], l1); memcpy(B.seq.p + o2, b2.seq.p + b2.off[p], l2);
            memcpy(B.qual.p + o1, b1.qual.p + b1.off[p], l1); memcpy(B.qual.p + o2, b2.qual.p + b2.off[p], l2);
            memcpy(B.names.p + m1, b1.names.p + b1.name_off[p], (size_t)(b1.name_off[p + 1] - b1.name_off[p]));
            memcpy(B.names.p + m2, b2.names.p + b2.name_off[p], (size_t)(b2.name_off[p + 1] - b2.name_off[p]));
        }
    };
    { std::vector<std::thread> th; for (int t = 1; t < P; ++t) th.emplace_back(part, n * t / P, n * (t + 1) / P); part(0, n / P); for (auto& t : th) t.join(); }
    B.off[2 * n] = b1.off[n] + b2.off[n]; B.name_off[2 * n] = b1.name_off[n] + b2.name_off[n];
    return n;
}

}  // namespace fastpath

// ---- paired-end (-1 / -2): st_align's paired loop (align_reads_dispatcher.hpp:356-389) over the C ABI ----------------------------------
// Learn the insert-size model on batches of -b pairs until it is complete (or the input ends), align those batches, then the rest (in
// larger batches: with the model fixed every pair is independent).  -u switches orphan recovery off.
struct AnyReader {
    MappedReader m; Reader* z = nullptr; bool mapped = false;
    explicit AnyReader(const std::string& path) { mapped = m.open(path); if (!mapped) z = new Reader(path); }
    ~AnyReader() { delete z; }
    bool next(Batch& b) { return mapped ? m.next(b) : z->next(b); }
};
static size_t read_pairs(AnyReader& r1, AnyReader& r2, size_t n_pairs, Batch& b) {
    size_t n = 0;
    while (n < n_pairs) {
        const bool g1 = r1.next(b);
        if (!g1) { Batch t; if (r2.next(t)) die("the mate files have different numbers of records"); break; }
        if (!r2.next(b)) die("the mate files have different numbers of records");
        ++n;
    }
    return n;
}
// the same with the two files parsed side by side (one thread each) and interleaved afterwards (mates 2p, 2p + 1), by T threads
static size_t read_pairs_par(AnyReader& r1, AnyReader& r2, size_t n_pairs, Batch& b, int T) {
    Batch b1, b2;
    size_t n2 = 0;
    std::thread t2([&] { while (n2 < n_pairs && r2.next(b2)) ++n2; });
    size_t n1 = 0;
    while (n1 < n_pairs && r1.next(b1)) ++n1;
    t2.join();
    if (n1 != n2) die("the mate files have different numbers of records");
    if (n1 < n_pairs) { Batch t; if (r1.next(t) || r2.next(t)) die("the mate files have different numbers of records"); }
    const size_t n = n1;
    if (!n) return 0;
    b.has_qual = b1.has_qual && b2.has_qual;
    b.off.resize(2 * n + 1); b.name_off.resize(2 * n + 1);
    for (size_t p = 0; p < n; ++p) {
        b.off[2 * p] = b1.off[p] + b2.off[p]; b.off[2 * p + 1] = b1.off[p + 1] + b2.off[p];
        b.name_off[2 * p] = b1.name_off[p] + b2.name_off[p]; b.name_off[2 * p + 1] = b1.name_off[p + 1] + b2.name_off[p];
    }
    b.off[2 * n] = b1.off[n] + b2.off[n]; b.name_off[2 * n] = b1.name_off[n] + b2.name_off[n];
    b.seq.resize(b.off[2 * n]); b.names.resize(b.name_off[2 * n]);
    if (b.has_qual) b.qual.resize(b.off[2 * n]);
    auto part = [&](size_t lo, size_t hi) {
        for (size_t p = lo; p < hi; ++p) {
            const size_t l1 = (size_t)(b1.off[p + 1] - b1.off[p]), l2 = (size_t)(b2.off[p + 1] - b2.off[p]);
            memcpy(b.seq.data() + b.off[2 * p], b1.seq.data() + b1.off[p], l1); memcpy(b.seq.data() + b.off[2 * p + 1], b2.seq.data() + b2.off[p], l2);
            if (b.has_qual) { memcpy(b.qual.data() + b.off[2 * p], b1.qual.data() + b1.off[p], l1); memcpy(b.qual.data() + b.off[2 * p + 1], b2.qual.data() + b2.off[p], l2); }
            memcpy(b.names.data() + b.name_off[2 * p], b1.names.data() + b1.name_off[p], (size_t)(b1.name_off[p + 1] - b1.name_off[p]));
            memcpy(b.names.data() + b.name_off[2 * p + 1], b2.names.data() + b2.name_off[p], (size_t)(b2.name_off[p + 1] - b2.name_off[p]));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(part, n * t / T, n * (t + 1) / T);
    part(0, n / T);
    for (auto& t : th) t.join();
    return n;
}
static int run_paired(Args& a, const std::string& sam_filename) {
    a.PE.find_orphan = a.find_orphan ? 1 : 0;            // -u switches orphan recovery off (align_full_ksw2.cpp:248-250)
    a.PE.secondary_chains = a.secondary ? 1 : 0;         // -Z: find_chains_secondary (the second track of the chaining, on the staged paired kernels)
    info("Output file: " + sam_filename);
    AnyReader r1(a.mate1), r2(a.mate2);
    if (a.dry_run) {
        Batch b;
        const size_t n = read_pairs(r1, r2, (size_t)-1, b);
        printf("dry-run: pairs=%zu bases=%zu min_len=%u filter_dir=%u dir_thr=%.1f find_orphan=%u b=%zu threads=%zu gpus=%d out=%s first=%.*s second=%.*s\n", n, b.seq.size(),
               a.P.min_len, a.PE.filter_dir, a.PE.dir_thr, a.PE.find_orphan, a.b, a.th, a.gpus, sam_filename.c_str(), n ? (int)b.name_off[1] : 0,
               n ? (const char*)b.names.data() : "", n ? (int)(b.name_off[2] - b.name_off[1]) : 0, n ? (const char*)b.names.data() + b.name_off[1] : "");
        return 0;
    }
    const std::string idx_path = a.filename + ".mfi";
    const bool have_mfi = !a.no_lcp && access(idx_path.c_str(), R_OK) == 0;          // (the flat file always carries the LCP samples)
    const std::string ms_path = a.filename + (a.no_lcp ? ".thrbv.full.ms" : ".thrbv.full.lcp.ms"), ldx_path = a.filename + ".ldx", txt_path = a.filename + ".txt";
    const bool have_txt = access(txt_path.c_str(), R_OK) == 0;          // optional: without it the text is rebuilt from the BWT on the GPU
    std::vector<moni_index_t*> idx(a.gpus, nullptr);
    std::vector<moni_ctx_t*> ctx(a.gpus, nullptr);
    for (int g = 0; g < a.gpus; ++g) {
        if (have_mfi) { if (moni_index_load(idx_path.c_str(), g, &idx[g])) die("cannot load " + idx_path + " on GPU " + std::to_string(g) + " (moni-hip has no CPU path)"); }
        else if (moni_index_load_reference(ms_path.c_str(), ldx_path.c_str(), have_txt ? txt_path.c_str() : nullptr, g, &idx[g]))
            die("cannot load " + idx_path + " nor " + ms_path + " + " + ldx_path + " (reader of the reference's files: layout unverified against a real `moni build` output) on GPU " + std::to_string(g) + " (moni-hip has no CPU path)");
        if (moni_ctx_create(idx[g], &ctx[g])) die("cannot create a context on GPU " + std::to_string(g));
    }
    const auto t0 = std::chrono::steady_clock::now();
    FILE* out = fopen(sam_filename.c_str(), "w");
    if (!out) die("open() file " + sam_filename + " failed");
    { char* h; uint64_t hl; if (moni_sam_header(idx[0], &h, &hl)) die("header"); put(h, hl, out); moni_free(h); }
    // -c: <sam>.csv, one line per pair (aligner_ksw2.hpp:911-914; header of aligner::to_csv, :3230-3234); the blocks of the batches are kept and written in input order at the end
    FILE* out2 = nullptr;
    std::map<size_t, std::string> csv_blocks;
    std::mutex mu_csv;
    std::string csv_first;
    if (a.csv) {
        out2 = fopen((sam_filename + ".csv").c_str(), "w");
        if (!out2) die("open() file " + sam_filename + ".csv failed");
        static const char hdr[] = "Read,Unique,Total,Max_Freq,Min_Freq,Highest_Occ,Lowest_Occ,Filtered,Chains_Skipped\n";
        put(hdr, sizeof hdr - 1, out2);
    }
    moni_pe_model_t model;
    memset(&model, 0, sizeof model);
    size_t processed = 0, aligned = 0;
    auto align_one = [&](int g, Batch& b, char** sam, uint64_t* len, uint64_t* n_al) {
        moni_read_batch_t rb{b.seq.data(), b.off.data(), b.n()};
        moni_align_stats_t st;
        if (a.report_mems) {           // -m: the MEM records instead of the pair's (the fragment model plays no part in them)
            const int rm = moni_pe_report_mems_batch(ctx[g], &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &a.PE, sam, len);
            if (rm) die("moni_pe_report_mems_batch failed (" + std::to_string(rm) + ")");
            *n_al = 0;
            return;
        }
        if (a.csv) {
            char* cs = nullptr; uint64_t cl = 0;
            const int rc = moni_pe_align_csv_batch(ctx[g], &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &a.PE, &model, sam, len, &cs, &cl, &st);
            if (rc) die("moni_pe_align_csv_batch failed (" + std::to_string(rc) + ")");
            csv_first.assign(cs, cl); moni_free(cs);
            *n_al = st.aligned;
            return;
        }
        const int rc = moni_pe_align_batch(ctx[g], &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &a.PE, &model, sam, len, &st);
        if (rc) die("moni_pe_align_batch failed (" + std::to_string(rc) + (rc == MONI_ERANGE ? ": a pair exceeds the paired path's capacities)" : ")"));
        *n_al = st.aligned;
    };
    std::vector<Batch*> learnt;
    while (!model.complete && !a.report_mems) {
        Batch* b = new Batch();
        if (!read_pairs(r1, r2, a.b, *b)) { delete b; break; }
        moni_read_batch_t rb{b->seq.data(), b->off.data(), b->n()};
        const int rc = moni_pe_learn_batch(ctx[0], &rb, &a.P, &a.PE, &model);
        if (rc) die("moni_pe_learn_batch failed (" + std::to_string(rc) + ")");
        learnt.push_back(b);
    }
    info("Insert size model: count " + std::to_string(model.count) + ", mean " + std::to_string(model.mean) + ", std dev " + std::to_string(model.std_dev) +
         (model.complete ? "" : " (input ended before " + std::to_string(a.PE.ins_learning_n) + " pairs)"));
    {   // the learning batches as one batch
        Batch all;
        for (Batch* b : learnt) {
            const uint64_t s0 = all.seq.size(), n0 = all.names.size();
            all.seq.insert(all.seq.end(), b->seq.begin(), b->seq.end()); all.qual.insert(all.qual.end(), b->qual.begin(), b->qual.end());
            all.names.insert(all.names.end(), b->names.begin(), b->names.end());
            for (size_t i = 1; i < b->off.size(); ++i) { all.off.push_back(s0 + b->off[i]); all.name_off.push_back(n0 + b->name_off[i]); }
            all.has_qual = all.has_qual && b->has_qual;
            delete b;
        }
        if (all.n()) {
            char* sam = nullptr; uint64_t len = 0, n_al = 0;
            align_one(0, all, &sam, &len, &n_al);
            put(sam, len, out); moni_free(sam);
            if (out2) put(csv_first.data(), csv_first.size(), out2);
            processed += all.n() / 2; aligned += n_al;
        }
    }
    // ---- the rest: reader thread (the two files parsed side by side) -> bounded queue -> a few workers per GPU, each with its own context (one's upload and
    // seeding run beside the others' paired kernels) -> every block written at its place in the file (the in-order prefix sum of the blocks' lengths) ----
    const size_t pairs_per_batch = std::max<size_t>(a.gpu_batch / 2, 1024);          // as many mates per batch as the single-end path has reads
    const int per_gpu = a.report_mems ? 1 : std::max(1, a.ctx_per_gpu), P = 4;          // contexts per GPU (--ctx-per-gpu, 3): one's upload, seeding and file write beside the others' paired kernels
    std::vector<moni_ctx_t*> wctx((size_t)a.gpus * per_gpu, nullptr);
    for (int g = 0; g < a.gpus; ++g) for (int k = 0; k < per_gpu; ++k) {
        if (k == 0) wctx[(size_t)g * per_gpu] = ctx[g];
        else if (moni_ctx_create(idx[g], &wctx[(size_t)g * per_gpu + k])) die("cannot create a context on GPU " + std::to_string(g));
    }
    if (fflush(out) != 0) die("short write to the output file");
    const uint64_t off0 = (uint64_t)ftello(out);
    const int fd = fileno(out);
    // a batch for the workers: the slow reader's Batch, or the fast reader's PairBatch (both mate files mapped four-line FASTQ)
    struct Item { size_t id; Batch* b; fastpath::PairBatch* f; };
    const bool fast_reader = r1.mapped && r2.mapped && r1.m.p[r1.m.at < r1.m.n ? r1.m.at : 0] == '@' && r2.m.p[r2.m.at < r2.m.n ? r2.m.at : 0] == '@' && getenv("MONI_CLI_SLOW_READER") == nullptr;
    std::mutex mu_q, mu_o;
    std::condition_variable cv_put, cv_get, cv_o;
    std::deque<Item> queue;
    bool reader_done = false;
    const size_t q_cap = wctx.size() + 1;
    std::map<size_t, uint64_t> lens, starts;
    size_t upto = 0; uint64_t off_upto = off0;
    std::atomic<size_t> n_proc{0}, n_al_all{0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_read = 0; std::vector<double> t_lib(wctx.size(), 0), t_wait(wctx.size(), 0), t_write(wctx.size(), 0);
    const bool verbose_batches = getenv("MONI_CLI_VERBOSE") != nullptr;
    std::thread reader([&] {
        size_t id = 0;
        const char* fbase[2] = {r1.m.p, r2.m.p}; const char* fend[2] = {r1.m.p + r1.m.n, r2.m.p + r2.m.n}; size_t fat[2] = {r1.m.at, r2.m.at};
        fastpath::ParsedBatch pb[2];
        while (true) {
            Batch* b = nullptr; fastpath::PairBatch* f = nullptr;
            const double r0 = now();
            if (fast_reader) { f = new fastpath::PairBatch(); if (!fastpath::read_pairs_fast(fbase, fend, fat, pairs_per_batch, 2 * P, pb, *f)) { delete f; break; } }
            else { b = new Batch(); if (!read_pairs_par(r1, r2, pairs_per_batch, *b, P)) { delete b; break; } }
            t_read += now() - r0;
            std::unique_lock<std::mutex> lk(mu_q);
            cv_put.wait(lk, [&] { return queue.size() < q_cap; });
            queue.push_back(Item{id++, b, f});
            cv_get.notify_one();
        }
        std::lock_guard<std::mutex> lk(mu_q);
        reader_done = true;
        cv_get.notify_all();
    });
    auto worker = [&](int w) {
        moni_ctx_t* C = wctx[w];
        while (true) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu_q);
                cv_get.wait(lk, [&] { return !queue.empty() || reader_done; });
                if (queue.empty()) return;
                it = queue.front(); queue.pop_front();
                cv_put.notify_one();
            }
            const uint8_t* v_seq = it.f ? it.f->seq.p : it.b->seq.data(); const uint64_t* v_off = it.f ? it.f->off.data() : it.b->off.data();
            const uint8_t* v_names = it.f ? it.f->names.p : it.b->names.data(); const uint64_t* v_noff = it.f ? it.f->name_off.data() : it.b->name_off.data();
            const uint8_t* v_qual = it.f ? it.f->qual.p : (it.b->has_qual ? it.b->qual.data() : nullptr);
            const size_t v_n = it.f ? it.f->n_reads : it.b->n();
            moni_read_batch_t rb{v_seq, v_off, v_n};
            char* sam = nullptr; uint64_t len = 0; size_t n_al = 0;
            const double x1 = now();
            if (a.report_mems) {
                const int rm = moni_pe_report_mems_batch(C, &rb, v_names, v_noff, v_qual, &a.P, &a.PE, &sam, &len);
                if (rm) die("moni_pe_report_mems_batch failed (" + std::to_string(rm) + ")");
            } else if (a.csv) {
                moni_align_stats_t st;
                char* cs = nullptr; uint64_t cl = 0;
                const int rc = moni_pe_align_csv_batch(C, &rb, v_names, v_noff, v_qual, &a.P, &a.PE, &model, &sam, &len, &cs, &cl, &st);
                if (rc) die("moni_pe_align_csv_batch failed (" + std::to_string(rc) + ")");
                { std::lock_guard<std::mutex> lk(mu_csv); csv_blocks[it.id].assign(cs, cl); }
                moni_free(cs);
                n_al = st.aligned;
            } else {          // the text in the context's pinned buffer: written out before the context's next call
                moni_align_stats_t st;
                const int rc = moni_pe_align_stream(C, &rb, v_names, v_noff, v_qual, &a.P, &a.PE, &model, &sam, &len, &st);
                if (rc) die("moni_pe_align_stream failed (" + std::to_string(rc) + (rc == MONI_ERANGE ? ": a pair exceeds the paired path's capacities)" : ")"));
                n_al = st.aligned;
            }
            uint64_t at = 0;
            const double x2 = now();
            {
                std::unique_lock<std::mutex> lk(mu_o);
                lens[it.id] = len;
                while (lens.count(upto)) { starts[upto] = off_upto; off_upto += lens[upto]; lens.erase(upto); ++upto; }
                cv_o.notify_all();
                cv_o.wait(lk, [&] { return upto > it.id; });
                at = starts[it.id]; starts.erase(it.id);
            }
            const double x3 = now();
            if (len) {
                std::vector<std::thread> th;
                const size_t sl = (len + P - 1) / P;
                for (int i = 1; i < P; ++i) { const size_t lo = std::min<size_t>(len, sl * i), hi = std::min<size_t>(len, sl * (i + 1)); if (hi > lo) th.emplace_back(fastpath::pwrite_all, fd, sam + lo, hi - lo, at + lo); }
                fastpath::pwrite_all(fd, sam, std::min<size_t>(len, sl), at);
                for (auto& t : th) t.join();
            }
            const double x4 = now();
            t_lib[w] += x2 - x1; t_wait[w] += x3 - x2; t_write[w] += x4 - x3;
            if (verbose_batches) fprintf(stderr, "batch %zu (worker %d, %zu pairs): library %.0f ms, wait %.0f ms, write %.0f ms, done at %.3f s\n", it.id, w, v_n / 2, (x2 - x1) * 1e3, (x3 - x2) * 1e3, (x4 - x3) * 1e3,
                                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (a.report_mems || a.csv) moni_free(sam);
            n_proc += v_n / 2; n_al_all += n_al;
            delete it.b; delete it.f;
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t w = 0; w < wctx.size(); ++w) th.emplace_back(worker, (int)w);
        for (auto& t : th) t.join();
    }
    reader.join();
    if (out2) { for (auto& kv : csv_blocks) put(kv.second.data(), kv.second.size(), out2); if (fclose(out2) != 0) die("short write to the csv file"); }
    processed += n_proc.load(); aligned += n_al_all.load();
    {
        double sl = 0, sw = 0, sr = 0;
        for (size_t w = 0; w < wctx.size(); ++w) { sl += t_lib[w]; sw += t_wait[w]; sr += t_write[w]; }
        info("Stage seconds: reading + interleaving the mate files " + std::to_string(t_read) + " (one reader thread, two parsers); summed over " + std::to_string(wctx.size()) + " workers: library calls " +
             std::to_string(sl) + ", waiting for the block's place " + std::to_string(sw) + ", file writes " + std::to_string(sr));
    }
    if (fseeko(out, (off_t)off_upto, SEEK_SET) != 0) die("seek in the output file failed");
    for (int g = 0; g < a.gpus; ++g) for (int k = 1; k < per_gpu; ++k) moni_ctx_destroy(wctx[(size_t)g * per_gpu + k]);
    close_out(out);
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info("Number of aligned pairs: " + std::to_string(aligned) + "/" + std::to_string(processed));
    info("Elapsed time (s): " + std::to_string(el));
    info("Pairs per second: " + std::to_string(processed / (el > 0 ? el : 1)));
    for (int g = 0; g < a.gpus; ++g) { moni_ctx_destroy(ctx[g]); moni_index_destroy(idx[g]); }
    return 0;
}

int main(int argc, char** argv) {
    // HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues; every context runs its launches on several streams (the paired
    // path on ~11), and streams that share a queue run one after the other
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    Args a;
    parse(argc, argv, a);
    // several contexts per GPU keep it busy by themselves: the library's sub-batches can be larger then (fewer, longer launches; bench.py --inflight, profiles/r04o)
    if (a.ctx_per_gpu >= 2) setenv("MONI_ALIGN_SUB", "500000", 0);
    // -Z (secondary chains) acts in the paired path only, as in the reference (aligner_ksw2.hpp:1190-1191): run_paired passes it on; single-end input ignores it
    // -n: <prefix>.thrbv.full.ms (ms_pointers<>: no LCP samples; the occurrence walks measure the LCP on the text, seed_finder.hpp:346-370).
    // -q: the reference would take the text from <prefix>.slp (SelfShapedSlp) instead of <prefix>.plain.slp; both grammars spell the same
    //     text and neither is read here - the text is rebuilt from the BWT - so the flag changes nothing (align_full_ksw2.cpp:414-426)
    const bool paired = !a.mate1.empty() || !a.mate2.empty();
    if (paired && (a.mate1.empty() || a.mate2.empty())) die("paired-end alignment needs both -1 and -2");
    if (paired && (a.legacy_ms || a.legacy_mems)) die("--ms / --mems take single-end input (-p)");
    if (!paired && a.patterns.empty()) die("no reads given (-p)");
    std::string fn = a.filename;
    std::vector<char> tmp(fn.begin(), fn.end()); tmp.push_back(0);
    const std::string base_name = basename(tmp.data());
    std::string sam_filename = (paired ? a.mate1 : a.patterns) + "_" + base_name + "_" + std::to_string(a.P.min_len) + ".sam";   // align_full_ksw2.cpp:347-353
    if (!a.output.empty()) sam_filename = a.output;
    if (paired) return run_paired(a, sam_filename);
    const bool legacy = a.legacy_ms || a.legacy_mems;
    if (legacy && a.output.empty()) sam_filename = a.patterns + "_" + base_name;       // mems.cpp / matching_statistics.cpp: <patterns>_<index> + .mems / .pointers / .lengths
    info("Output file: " + sam_filename);
    MappedReader mrd;
    const bool mapped = mrd.open(a.patterns);
    Reader* zrd = mapped ? nullptr : new Reader(a.patterns);
    auto next_record = [&](Batch& b) { return mapped ? mrd.next(b) : zrd->next(b); };
    if (a.dry_run) {
        Batch b; size_t n = 0, bases = 0;
        while (next_record(b)) { ++n; }
        bases = b.seq.size();
        printf("dry-run: reads=%zu bases=%zu min_len=%u ext_len=%u S=%u F=%.2f O=%d,%d E=%d,%d threads=%zu gpus=%d out=%s first=%.*s\n", n, bases, a.P.min_len,
               a.P.ext_len, a.P.n_seeds_thr, a.P.freq_thr, a.P.gapo, a.P.gapo2, a.P.gape, a.P.gape2, a.th, a.gpus, sam_filename.c_str(),
               n ? (int)b.name_off[1] : 0, n ? (const char*)b.names.data() : "");
        if (!a.dry_write || !mapped) return 0;
    }
    const std::string idx_path = a.filename + ".mfi";
    const bool fast = mapped && !legacy && !a.report_mems && !a.csv && getenv("MONI_CLI_QUEUE_PATH") == nullptr;
    const int per_gpu = legacy ? 1 : (fast ? a.ctx_per_gpu : 2);                      // contexts (batches in flight) per GPU
    std::vector<moni_index_t*> idx(a.gpus, nullptr);
    std::vector<moni_ctx_t*> ctx((size_t)a.gpus * per_gpu, nullptr);
    const bool have_mfi = !a.no_lcp && access(idx_path.c_str(), R_OK) == 0;          // (the flat file always carries the LCP samples)
    const std::string ms_path = a.filename + (a.no_lcp ? ".thrbv.full.ms" : ".thrbv.full.lcp.ms"), ldx_path = a.filename + ".ldx", txt_path = a.filename + ".txt";
    const bool have_txt = access(txt_path.c_str(), R_OK) == 0;          // optional: without it the text is rebuilt from the BWT on the GPU
    for (int g = 0; g < a.gpus && !a.dry_write; ++g) {
        if (have_mfi) { if (moni_index_load(idx_path.c_str(), g, &idx[g])) die("cannot load " + idx_path + " on GPU " + std::to_string(g) + " (moni-hip has no CPU path)"); }
        else if (moni_index_load_reference(ms_path.c_str(), ldx_path.c_str(), have_txt ? txt_path.c_str() : nullptr, g, &idx[g]))
            die("cannot load " + idx_path + " nor " + ms_path + " + " + ldx_path + " (reader of the reference's files: layout unverified against a real `moni build` output) on GPU " + std::to_string(g) + " (moni-hip has no CPU path)");
        for (int k = 0; k < per_gpu; ++k) if (moni_ctx_create(idx[g], &ctx[(size_t)g * per_gpu + k])) die("cannot create a context on GPU " + std::to_string(g));
    }
    if (fast) {
        using namespace fastpath;
        const auto t0 = std::chrono::steady_clock::now();
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const char* base = mrd.p; const char* end = mrd.p + mrd.n;
        const bool fq = base[0] == '@';
        // record-aligned ranges of about gpu_batch reads (the first record's size is the estimate)
        const char* r1 = base; for (int k = 0; k < (fq ? 4 : 2); ++k) r1 = next_line(r1, end);
        const size_t rec_bytes = std::max<size_t>((size_t)(r1 - base), 8), chunk_bytes = std::max<size_t>(rec_bytes * a.gpu_batch, 1 << 16);
        std::vector<const char*> cuts(1, base);
        while (cuts.back() < end) { const char* nx = (size_t)(end - cuts.back()) <= chunk_bytes + chunk_bytes / 4 ? end : align_record(cuts.back() + chunk_bytes, base, end, fq); if (nx <= cuts.back()) nx = end; cuts.push_back(nx); }
        const size_t n_chunks = cuts.size() - 1;
        const int fd = ::open(sam_filename.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("open() file " + sam_filename + " failed");
        uint64_t hdr_len = 0;
        if (!a.dry_write) { char* h; uint64_t hl; if (moni_sam_header(idx[0], &h, &hl)) die("header"); pwrite_all(fd, h, hl, 0); hdr_len = hl; moni_free(h); }
        std::mutex mu; std::condition_variable cv;
        std::vector<uint64_t> lens(n_chunks, ~0ull), starts(n_chunks, 0);
        size_t upto = 0; uint64_t off_upto = hdr_len;
        std::atomic<size_t> next_chunk{0};
        std::atomic<size_t> processed{0}, aligned{0};
        const int P = (int)std::max<size_t>(1, std::min<size_t>(8, a.th / std::max<size_t>(1, ctx.size() / (size_t)a.gpus)));      // helper threads of one worker
        std::vector<double> t_parse(ctx.size(), 0), t_lib(ctx.size(), 0), t_wait(ctx.size(), 0), t_write(ctx.size(), 0);
        const bool verbose_ranges = getenv("MONI_CLI_VERBOSE") != nullptr;
        auto worker = [&](int w) {
            ParsedBatch B;
            while (true) {
                const size_t id = next_chunk.fetch_add(1);
                if (id >= n_chunks) return;
                double x0 = now();
                parse_range(cuts[id], cuts[id + 1], base, end, fq, P, B);
                double x1 = now(); t_parse[w] += x1 - x0;
                char* sam = nullptr; uint64_t len = 0;
                moni_align_stats_t st; memset(&st, 0, sizeof st);
                std::string placeholder;
                if (a.dry_write) {          // one unaligned record per read (what the library returns for a read without seeds), worker id in a tag
                    for (size_t r = 0; r < B.n; ++r) {
                        placeholder.append((const char*)B.names.p + B.name_off[r], (size_t)(B.name_off[r + 1] - B.name_off[r]));
                        placeholder += "\t4\t*\t0\t255\t*\t*\t0\t0\t";
                        placeholder.append((const char*)B.seq.p + B.off[r], (size_t)(B.off[r + 1] - B.off[r]));
                        placeholder += "\t";
                        if (fq) placeholder.append((const char*)B.qual.p + B.off[r], (size_t)(B.off[r + 1] - B.off[r])); else placeholder += "*";
                        placeholder += "\tXW:i:" + std::to_string(w) + "\n";
                    }
                    sam = placeholder.data(); len = placeholder.size();
                    std::this_thread::sleep_for(std::chrono::milliseconds(3));          // (a library call takes longer than that: the other workers get their ranges)
                } else if (B.n) {
                    moni_read_batch_t rb{B.seq.p, B.off.data(), (uint64_t)B.n};
                    if (moni_align_stream(ctx[w], &rb, B.names.p, B.name_off.data(), fq ? B.qual.p : nullptr, &a.P, &sam, &len, &st)) die("moni_align_stream failed");
                }
                double x2 = now(); t_lib[w] += x2 - x1;
                uint64_t at;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    lens[id] = len;
                    while (upto < n_chunks && lens[upto] != ~0ull) { starts[upto] = off_upto; off_upto += lens[upto]; ++upto; }
                    cv.notify_all();
                    cv.wait(lk, [&] { return upto > id; });
                    at = starts[id];
                }
                double x3 = now(); t_wait[w] += x3 - x2;
                if (len) {
                    std::vector<std::thread> th;
                    const size_t sl = (len + P - 1) / P;
                    for (int i = 1; i < P; ++i) { const size_t lo = std::min<size_t>(len, sl * i), hi = std::min<size_t>(len, sl * (i + 1)); if (hi > lo) th.emplace_back(pwrite_all, fd, sam + lo, hi - lo, at + lo); }
                    pwrite_all(fd, sam, std::min<size_t>(len, sl), at);
                    for (auto& t : th) t.join();
                }
                const double x4 = now();
                t_write[w] += x4 - x3;
                if (verbose_ranges) fprintf(stderr, "range %zu (worker %d, %zu reads): parse %.0f ms, library %.0f ms (seed %.0f, align kernels %.0f), wait %.0f ms, write %.0f ms, done at %.3f s\n", id, w, B.n,
                                            (x1 - x0) * 1e3, (x2 - x1) * 1e3, st.t_seed * 1e3, st.t_dp_kernel * 1e3, (x3 - x2) * 1e3, (x4 - x3) * 1e3,
                                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                processed += B.n; aligned += st.aligned;
            }
        };
        std::vector<std::thread> th;
        for (size_t w = 0; w < ctx.size(); ++w) th.emplace_back(worker, (int)w);
        for (auto& t : th) t.join();
        if (::close(fd) != 0) die("close() of the output file failed");
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        info("Number of aligned reads: " + std::to_string(aligned.load()) + "/" + std::to_string(processed.load()));
        info("Elapsed time (s): " + std::to_string(el));
        info("Reads per second: " + std::to_string(processed.load() / (el > 0 ? el : 1)));
        double sp = 0, sl2 = 0, sw = 0, sr = 0;
        for (size_t w = 0; w < ctx.size(); ++w) { sp += t_parse[w]; sl2 += t_lib[w]; sw += t_wait[w]; sr += t_write[w]; }
        info("Stage seconds summed over " + std::to_string(ctx.size()) + " workers (" + std::to_string(n_chunks) + " ranges, " + std::to_string(P) + " helper threads each): parse " + std::to_string(sp) +
             ", library calls " + std::to_string(sl2) + ", waiting for the block's place " + std::to_string(sw) + ", file writes " + std::to_string(sr));
        for (size_t w = 0; w < ctx.size(); ++w) if (ctx[w]) moni_ctx_destroy(ctx[w]);
        for (int g = 0; g < a.gpus; ++g) if (idx[g]) moni_index_destroy(idx[g]);
        return 0;
    }
    if (a.dry_write) return 0;
    FILE* out = nullptr; FILE* out2 = nullptr;
    if (a.legacy_ms) { out = fopen((sam_filename + ".pointers").c_str(), "w"); out2 = fopen((sam_filename + ".lengths").c_str(), "w"); if (!out || !out2) die("open() file " + sam_filename + ".pointers/.lengths failed"); }
    else if (a.legacy_mems) { out = fopen((sam_filename + ".mems").c_str(), "w"); if (!out) die("open() file " + sam_filename + ".mems failed"); }
    else {
        out = fopen(sam_filename.c_str(), "w");
        if (!out) die("open() file " + sam_filename + " failed");
        char* h; uint64_t hl; if (moni_sam_header(idx[0], &h, &hl)) die("header"); put(h, hl, out); moni_free(h);
        if (a.csv) {          // -c: <sam>.csv (align_reads_dispatcher.hpp:302,334-339), header of aligner::to_csv (aligner_ksw2.hpp:3230-3234)
            out2 = fopen((sam_filename + ".csv").c_str(), "w");
            if (!out2) die("open() file " + sam_filename + ".csv failed");
            static const char hdr[] = "Read,Unique,Total,Max_Freq,Min_Freq,Highest_Occ,Lowest_Occ,Filtered,Chains_Skipped\n";
            put(hdr, sizeof hdr - 1, out2);
        }
    }
    auto t0 = std::chrono::steady_clock::now();
    // ---- reader thread -> bounded queue of parsed batches -> workers -> bounded in-order window -> writer thread ----
    struct Item { size_t id; Batch* b; };
    struct Done { char* a = nullptr; uint64_t la = 0; std::string b; };      // a: malloc'ed block of the library; b: a second stream (.lengths)
    std::mutex mu_q, mu_out;
    std::condition_variable cv_q_put, cv_q_get, cv_out, cv_window;
    std::deque<Item> queue;
    bool reader_done = false;
    const size_t q_cap = (size_t)a.gpus * per_gpu + 2, window = 2 * (size_t)a.gpus * per_gpu + 2;
    size_t next_out = 0, n_batches = 0, processed = 0, aligned = 0;
    double t_reader = 0, t_writer = 0, t_align = 0, t_wait_window = 0;      // busy seconds of the stages (t_align summed over the workers)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    bool workers_done = false;
    std::map<size_t, Done> done;
    std::thread reader([&]() {
        size_t id = 0;
        while (true) {
            const double r0 = now();
            Batch* b = new Batch();
            b->seq.reserve(a.gpu_batch * 160); b->qual.reserve(a.gpu_batch * 160); b->names.reserve(a.gpu_batch * 24);
            // the matching-statistics workspace of a batch is strided by its longest read (2 x reads x longest x 8 bytes): the legacy modes,
            // which take long patterns, close a batch when that product passes 2^28 entries (4 GB)
            size_t longest = 0;
            while (b->n() < a.gpu_batch && next_record(*b)) {
                if (!legacy) continue;
                longest = std::max(longest, (size_t)(b->off[b->n()] - b->off[b->n() - 1]));
                if (b->n() * longest > ((size_t)1 << 28)) break;
            }
            t_reader += now() - r0;
            if (b->n() == 0) { delete b; break; }
            std::unique_lock<std::mutex> lk(mu_q);
            cv_q_put.wait(lk, [&] { return queue.size() < q_cap; });
            queue.push_back(Item{id++, b});
            cv_q_get.notify_one();
        }
        std::lock_guard<std::mutex> lk(mu_q);
        reader_done = true; n_batches = id;
        cv_q_get.notify_all();
    });
    std::thread writer([&]() {
        while (true) {
            Done d;
            {
                std::unique_lock<std::mutex> lk(mu_out);
                cv_out.wait(lk, [&] { return (!done.empty() && done.begin()->first == next_out) || workers_done; });
                if (done.empty() || done.begin()->first != next_out) { if (workers_done && done.empty()) return; if (workers_done) die("a batch is missing from the output"); continue; }
                d = std::move(done.begin()->second);
                done.erase(done.begin());
                ++next_out;
                cv_window.notify_all();
            }
            const double w0 = now();
            if (d.la) put(d.a, d.la, out);
            if (d.a) moni_free(d.a);
            if (out2 && !d.b.empty()) put(d.b.data(), d.b.size(), out2);
            t_writer += now() - w0;
        }
    });
    auto worker = [&](int w) {
        moni_ctx_t* C = ctx[w];
        while (true) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu_q);
                cv_q_get.wait(lk, [&] { return !queue.empty() || reader_done; });
                if (queue.empty()) return;
                it = queue.front(); queue.pop_front();
                cv_q_put.notify_one();
            }
            Batch& b = *it.b;
            moni_read_batch_t rb{b.seq.data(), b.off.data(), b.n()};
            Done d;
            size_t n_al = 0;
            const double a0 = now();
            if (legacy) {
                std::vector<uint64_t> ptr(b.seq.size() + 1), len(b.seq.size() + 1);
                if (moni_ms_lengths_batch(C, &rb, ptr.data(), len.data())) die("moni_ms_lengths_batch failed");
                std::string sa, sb;
                for (size_t r = 0; r < b.n(); ++r) {
                    const std::string hdr = ">" + std::string((const char*)b.names.data() + b.name_off[r], (size_t)(b.name_off[r + 1] - b.name_off[r])) + "\n";
                    sa += hdr;
                    const size_t o = b.off[r], m = b.off[r + 1] - o;
                    if (a.legacy_ms) {          // matching_statistics.cpp:565-600: values followed by a blank, one line per read
                        sb += hdr;
                        for (size_t k = 0; k < m; ++k) { sa += std::to_string(ptr[o + k]); sa.push_back(' '); sb += std::to_string(len[o + k]); sb.push_back(' '); }
                        sb.push_back('\n');
                    } else {                    // mems.cpp:241-262, 585-590: (offset,length) of every position whose length does not drop
                        for (size_t k = 0; k < m; ++k) if (k == 0 || len[o + k] >= len[o + k - 1]) { sa += "(" + std::to_string(k) + "," + std::to_string(len[o + k]) + ") "; }
                    }
                    sa.push_back('\n');
                }
                d.a = (char*)malloc(sa.size() + 1); memcpy(d.a, sa.data(), sa.size()); d.la = sa.size(); d.b = std::move(sb);
            } else if (a.report_mems) {
                if (moni_report_mems_batch(C, &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &d.a, &d.la)) die("moni_report_mems_batch failed");
                n_al = b.n();
            } else if (a.csv) {
                moni_align_stats_t st; char* cs = nullptr; uint64_t cl = 0;
                if (moni_align_csv_batch(C, &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &d.a, &d.la, &cs, &cl, &st)) die("moni_align_csv_batch failed");
                d.b.assign(cs, cl); moni_free(cs);
                n_al = st.aligned;
            } else {
                moni_align_stats_t st;
                if (moni_align_batch(C, &rb, b.names.data(), b.name_off.data(), b.has_qual ? b.qual.data() : nullptr, &a.P, &d.a, &d.la, &st)) die("moni_align_batch failed");
                n_al = st.aligned;
            }
            const size_t n_reads = b.n();
            delete it.b;
            const double a1 = now();
            std::unique_lock<std::mutex> lk(mu_out);
            cv_window.wait(lk, [&] { return it.id < next_out + window; });
            t_align += a1 - a0; t_wait_window += now() - a1;      // bounded: a slow writer holds the workers back instead of the heap growing
            done[it.id] = std::move(d);
            processed += n_reads; aligned += n_al;
            cv_out.notify_one();
        }
    };
    std::vector<std::thread> th;
    for (size_t w = 0; w < ctx.size(); ++w) th.emplace_back(worker, (int)w);
    reader.join();
    for (auto& t : th) t.join();
    { std::lock_guard<std::mutex> lk(mu_out); workers_done = true; cv_out.notify_all(); }
    writer.join();
    close_out(out);
    close_out(out2);
    delete zrd;
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info("Number of aligned reads: " + std::to_string(aligned) + "/" + std::to_string(processed));
    info("Elapsed time (s): " + std::to_string(el));
    info("Reads per second: " + std::to_string(processed / (el > 0 ? el : 1)));
    info("Stage busy seconds: reader (parse) " + std::to_string(t_reader) + ", library calls summed over " + std::to_string(ctx.size()) + " workers " + std::to_string(t_align) +
         ", writer " + std::to_string(t_writer) + ", workers held back by the writer " + std::to_string(t_wait_window));
    for (size_t w = 0; w < ctx.size(); ++w) moni_ctx_destroy(ctx[w]);
    for (int g = 0; g < a.gpus; ++g) moni_index_destroy(idx[g]);
    return 0;
}
