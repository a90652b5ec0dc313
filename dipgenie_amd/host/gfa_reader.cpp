#include "gfa_reader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <tuple>
#include <unordered_map>

namespace dg {

namespace {

struct LineReader {
    gzFile fp = nullptr;
    std::vector<char> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    explicit LineReader(gzFile f) : fp(f), buf(1 << 22) {}
    // Reads one line (without '\n', trailing '\r' stripped like kseq's KS_SEP_LINE). False at EOF.
    bool next(std::string &line) {
        line.clear();
        bool got = false;
        for (;;) {
            if (beg == end) {
                if (eof) break;
                int n = gzread(fp, buf.data(), (unsigned)buf.size());
                if (n <= 0) { eof = true; break; }
                beg = 0; end = (size_t)n;
            }
            got = true;
            const char *p = (const char *)memchr(buf.data() + beg, '\n', end - beg);
            if (p) {
                line.append(buf.data() + beg, p - (buf.data() + beg));
                beg = (size_t)(p - buf.data()) + 1;
                break;
            }
            line.append(buf.data() + beg, end - beg);
            beg = end;
        }
        if (!got) return false;
        if (line.size() > 1 && line.back() == '\r') line.pop_back();
        return true;
    }
};

struct Arc { uint32_t v, w; int32_t ov, ow; };

struct Builder {
    GfaGraph &g;
    std::unordered_map<std::string, uint32_t> name2id;
    std::vector<Arc> arcs;
    explicit Builder(GfaGraph &gg) : g(gg) {}

    uint32_t add_seg(const std::string &name) {   // gfa-base.cpp:75-97
        auto it = name2id.find(name);
        if (it != name2id.end()) return it->second;
        uint32_t id = (uint32_t)g.seg_name.size();
        name2id.emplace(name, id);
        g.seg_name.push_back(name);
        g.seg_seq.emplace_back();
        g.seg_len.push_back(0);
        g.seg_del.push_back(0);
        return id;
    }
};

// split on tabs; returns pointers into a mutable copy
void split_tabs(std::string &s, std::vector<char *> &f) {
    f.clear();
    char *p = &s[0];
    f.push_back(p);
    for (size_t i = 0; i < s.size(); ++i)
        if (s[i] == '\t') { s[i] = 0; f.push_back(p + i + 1); }
}

void parse_S(Builder &b, std::vector<char *> &f) {          // gfa-io.cpp:214-277
    if (f.size() < 3) return;
    uint32_t id = b.add_seg(f[1]);
    int64_t LN = -1;
    for (size_t i = 3; i < f.size(); ++i)
        if (strncmp(f[i], "LN:i:", 5) == 0) LN = strtol(f[i] + 5, nullptr, 10);
    if (f[2][0] == '*' && f[2][1] == 0) {
        b.g.seg_seq[id].clear();
        b.g.seg_len[id] = LN >= 0 ? (uint32_t)LN : 0;
    } else {
        b.g.seg_seq[id] = f[2];
        b.g.seg_len[id] = (uint32_t)b.g.seg_seq[id].size();
    }
}

void parse_L(Builder &b, std::vector<char *> &f) {          // gfa-io.cpp:279-365
    if (f.size() < 5) return;
    if ((f[2][0] != '+' && f[2][0] != '-') || (f[4][0] != '+' && f[4][0] != '-')) return;
    int oriv = f[2][0] != '+', oriw = f[4][0] != '+';
    int32_t ov = 0, ow = 0;
    if (f.size() >= 6) {
        const char *q = f[5];
        if (*q == '*') {
            ov = ow = 0;
        } else if (*q == ':') {
            ov = INT32_MAX;
            ow = isdigit((unsigned char)q[1]) ? (int32_t)strtol(q + 1, nullptr, 10) : INT32_MAX;
        } else if (isdigit((unsigned char)*q)) {
            char *r;
            ov = (int32_t)strtol(q, &r, 10);
            if (isupper((unsigned char)*r)) {   // CIGAR
                ov = ow = 0;
                char *qq = const_cast<char *>(q);
                do {
                    long l = strtol(qq, &qq, 10);
                    if (*qq == 'M' || *qq == 'D' || *qq == 'N') ov += (int32_t)l;
                    if (*qq == 'M' || *qq == 'I' || *qq == 'S') ow += (int32_t)l;
                    ++qq;
                } while (isdigit((unsigned char)*qq));
            } else if (*r == ':') {
                ow = isdigit((unsigned char)r[1]) ? (int32_t)strtol(r + 1, nullptr, 10) : INT32_MAX;
            } else {
                return;   // invalid overlap field -> line rejected
            }
        } else {
            return;
        }
    }
    uint32_t v = b.add_seg(f[1]) << 1 | (uint32_t)oriv;
    uint32_t w = b.add_seg(f[3]) << 1 | (uint32_t)oriw;
    b.arcs.push_back({v, w, ov, ow});
}

// W lines are the bulk of a pangenome GFA (one step per segment per haplotype) and independent of each other, so
// they are kept as text while the file is read and parsed together afterwards, in parallel.  The reference resolves a
// step against the segments seen SO FAR (gfa-io.cpp:367-432): segment ids are handed out in first-appearance order,
// so "known when this line was read" is exactly id < n_seg_then.
struct PendingWalk { std::string line; uint32_t n_seg_then; };

// Segment names -> ids for the walk steps (10^8 lookups on a chr22-scale panel): open addressing over the entries of name2id, looked
// up by (pointer, length) -- no std::string per step, no node chasing.  Read-only; built once when the S lines are in.
struct NameIndex {
    std::vector<const std::pair<const std::string, uint32_t> *> slot, by_id;   // by_id: a walk mostly steps to one of the next few segments of the file
    uint64_t mask = 0;
    static uint64_t hash(const char *p, size_t n) {
        uint64_t h = 0xcbf29ce484222325ULL;
        for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 0x100000001b3ULL; }
        return h ^ (h >> 29);
    }
    explicit NameIndex(const std::unordered_map<std::string, uint32_t> &m) {
        size_t cap = 16;
        while (cap < 2 * m.size() + 2) cap <<= 1;
        slot.assign(cap, nullptr);
        mask = cap - 1;
        by_id.assign(m.size(), nullptr);
        for (const auto &kv : m) {
            uint64_t q = hash(kv.first.data(), kv.first.size()) & mask;
            while (slot[q]) q = (q + 1) & mask;
            slot[q] = &kv;
            if (kv.second < by_id.size()) by_id[kv.second] = &kv;
        }
    }
    // the same answer as find(), tried first on the ids right after `last` (sequential memory instead of two cache misses per step)
    const uint32_t *find_near(uint32_t last, const char *p, size_t n) const {
        for (uint32_t c = last + 1; c < last + 5 && c < by_id.size(); ++c) {
            const auto *e = by_id[c];
            if (e && e->first.size() == n && memcmp(e->first.data(), p, n) == 0) return &e->second;
        }
        return find(p, n);
    }
    const uint32_t *find(const char *p, size_t n) const {
        for (uint64_t q = hash(p, n) & mask; slot[q]; q = (q + 1) & mask)
            if (slot[q]->first.size() == n && memcmp(slot[q]->first.data(), p, n) == 0) return &slot[q]->second;
        return nullptr;
    }
};

void parse_W(const Builder &b, const NameIndex &names, PendingWalk &pw, GfaWalk &t, std::string &warnings) {
    std::vector<char *> f;
    split_tabs(pw.line, f);
    if (f.size() < 7) return;
    t.sample = f[1];
    t.hap = atoi(f[2]);
    const char *q = f[6];
    const char *end = q + strlen(q);
    const char *qq = q;
    (void)b;
    uint32_t last = 0xFFFFFFFFu;                                        // (+ 1 = 0: the first step tries the file's first segments)
    for (const char *pp = q + 1; pp <= end; ++pp) {
        if (pp == end || *pp == '>' || *pp == '<') {
            const uint32_t *id = names.find_near(last, qq + 1, (size_t)(pp - (qq + 1)));
            if (id) last = *id;
            if (id && *id < pw.n_seg_then) t.v.push_back(*id << 1 | (uint32_t)(*qq == '<'));
            else warnings += "WARNING: failed to find segment '" + std::string(qq + 1, pp - (qq + 1)) + "'\n";
            qq = pp;
        }
    }
}

void walk_flip(GfaGraph &g) {                               // gfa-io.cpp:64-93
    if (g.walks.empty()) return;
    std::vector<int8_t> strand(g.n_seg(), 0);
    for (auto &w : g.walks)
        for (uint32_t x : w.v)
            if (strand[x >> 1] == 0) strand[x >> 1] = (x & 1) ? -1 : 1;
    for (auto &w : g.walks) {
        int64_t n0 = 0, n1 = 0;
        for (uint32_t x : w.v) {
            int8_t s = (x & 1) ? -1 : 1;
            if (s == strand[x >> 1]) ++n0; else ++n1;
        }
        if (n0 >= n1) continue;
        size_t n = w.v.size();
        for (size_t j = 0; j < n >> 1; ++j) {
            uint32_t t = w.v[j] ^ 1;
            w.v[j] = w.v[n - 1 - j] ^ 1;
            w.v[n - 1 - j] = t;
        }
        if (n & 1) w.v[n >> 1] ^= 1;
    }
}

void finalize(Builder &b) {                                 // gfa-base.cpp:421-430 gfa_finalize
    GfaGraph &g = b.g;
    const uint32_t n = g.n_seg();
    for (uint32_t i = 0; i < n; ++i)                        // gfa_fix_no_seg
        if (g.seg_len[i] == 0) g.seg_del[i] = 1;
    // gfa_fix_symm_add: a link and its complement pair up one-to-one; unmatched ones get a
    // complement added. With counts n_T of type T=(v,w,ov,ow) and n_T' of T'=(w^1,v^1,ow,ov) the
    // final multiplicity of both is max(n_T,n_T') (n_T if T is its own complement).
    // (the distinct link types in sorted order with their counts: one sort instead of an ordered map of tuples)
    typedef std::tuple<uint32_t, uint32_t, int32_t, int32_t> Key;
    std::vector<Key> keys;
    keys.reserve(b.arcs.size());
    for (auto &a : b.arcs) keys.emplace_back(a.v, a.w, a.ov, a.ow);
    std::sort(keys.begin(), keys.end());
    std::vector<std::pair<Key, int64_t>> cnt;
    for (size_t q = 0; q < keys.size();) {
        size_t q1 = q;
        while (q1 < keys.size() && keys[q1] == keys[q]) ++q1;
        cnt.emplace_back(keys[q], (int64_t)(q1 - q));
        q = q1;
    }
    auto find = [&](const Key &k) -> const std::pair<Key, int64_t> * {
        auto it = std::lower_bound(cnt.begin(), cnt.end(), k, [](const std::pair<Key, int64_t> &x, const Key &y) { return x.first < y; });
        return it != cnt.end() && it->first == k ? &*it : nullptr;
    };
    std::vector<uint32_t> deg((size_t)2 * n, 0);
    for (int pass = 0; pass < 2; ++pass) {                  // count, then fill: no per-vertex reallocation
        if (pass == 1) { g.arcs.assign((size_t)2 * n, {}); for (size_t v = 0; v < deg.size(); ++v) if (deg[v]) g.arcs[v].reserve(deg[v]); }
        for (auto &kv : cnt) {
            uint32_t v = std::get<0>(kv.first), w = std::get<1>(kv.first);
            int32_t ov = std::get<2>(kv.first), ow = std::get<3>(kv.first);
            Key comp(w ^ 1, v ^ 1, ow, ov);
            int64_t m = kv.second;
            if (comp != kv.first) {
                const auto *it = find(comp);
                int64_t mc = it ? it->second : 0;
                if (mc > m) m = mc;
                if (!it) {   // complement absent: emit it here (it is not a key of cnt)
                    if (!g.seg_del[(w ^ 1) >> 1] && !g.seg_del[(v ^ 1) >> 1]) {
                        if (pass == 0) deg[w ^ 1] += (uint32_t)m; else for (int64_t c = 0; c < m; ++c) g.arcs[w ^ 1].push_back(v ^ 1);
                    }
                }
            }
            if (g.seg_del[v >> 1] || g.seg_del[w >> 1]) continue;   // gfa_fix_arc_len / gfa_arc_rm
            if (pass == 0) deg[v] += (uint32_t)m; else for (int64_t c = 0; c < m; ++c) g.arcs[v].push_back(w);
        }
    }
}

}  // namespace

bool read_gfa_file(const std::string &path, GfaGraph &g, std::string &err) {
    gzFile fp = path == "-" ? gzdopen(0, "r") : gzopen(path.c_str(), "r");
    if (!fp) { err = "cannot open " + path; return false; }
    gzbuffer(fp, 1 << 20);
    const bool dbg = getenv("DG_DEBUG") != nullptr;
    auto now = [] { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
    double tl = now();
    auto lap = [&](const char *w) { if (dbg) { double t = now(); fprintf(stderr, "[dg::gfa] %-18s %.3f s\n", w, t - tl); tl = t; } };
    Builder b(g);
    LineReader lr(fp);
    std::string line;
    std::vector<char *> f;
    std::vector<PendingWalk> pending;
    while (lr.next(line)) {
        if (line.size() < 3 || line[1] != '\t') continue;   // gfa-io.cpp:492
        char t = line[0];
        if (t != 'S' && t != 'L' && t != 'W') continue;
        if (t == 'W') {
            // (a W line with fewer than 7 fields adds no walk at all)
            if (std::count(line.begin(), line.end(), '\t') >= 6) { pending.push_back({std::string(), g.n_seg()}); pending.back().line.swap(line); }
            continue;
        }
        split_tabs(line, f);
        if (t == 'S') parse_S(b, f);
        else parse_L(b, f);
    }
    gzclose(fp);
    lap("read + parse S/L lines");
    {
        g.walks.resize(pending.size());
        std::vector<std::string> warn(pending.size());
        const NameIndex names(b.name2id);
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t w = 0; w < (int64_t)pending.size(); ++w) parse_W(b, names, pending[w], g.walks[w], warn[w]);
        for (auto &ws : warn) if (!ws.empty()) fputs(ws.c_str(), stderr);
    }
    lap("parse W lines");
    walk_flip(g);
    lap("walk_flip");
    finalize(b);
    lap("finalize arcs");
    return true;
}

}  // namespace dg
