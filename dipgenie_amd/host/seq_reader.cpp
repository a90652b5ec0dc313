#include "seq_reader.hpp"

#include <zlib.h>

#include <cctype>
#include <cstring>

namespace dg {

namespace {
struct Stream {
    gzFile fp;
    std::vector<unsigned char> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    explicit Stream(gzFile f) : fp(f), buf(1 << 20) {}
    int getc() {
        if (beg >= end) {
            if (eof) return -1;
            int n = gzread(fp, buf.data(), (unsigned)buf.size());
            if (n <= 0) { eof = true; return -1; }
            beg = 0; end = (size_t)n;
        }
        return buf[beg++];
    }
    // append the rest of the current line to s (newline consumed, trailing '\r' stripped)
    void rest_of_line(std::string *s) {
        size_t start = s ? s->size() : 0;
        for (;;) {
            if (beg >= end) {
                if (eof) break;
                int n = gzread(fp, buf.data(), (unsigned)buf.size());
                if (n <= 0) { eof = true; break; }
                beg = 0; end = (size_t)n;
            }
            unsigned char *p = (unsigned char *)memchr(buf.data() + beg, '\n', end - beg);
            size_t stop = p ? (size_t)(p - buf.data()) : end;
            if (s) s->append((const char *)buf.data() + beg, stop - beg);
            beg = p ? stop + 1 : end;
            if (p) break;
        }
        if (s && s->size() > start + 0 && s->size() > 1 && s->back() == '\r') s->pop_back();
    }
};
}  // namespace

bool read_sequences(const std::string &path, std::vector<std::pair<std::string, std::string>> &out,
                    std::string &err) {
    gzFile fp = gzopen(path.c_str(), "r");
    if (!fp) { err = "cannot open " + path; return false; }
    gzbuffer(fp, 1 << 20);
    Stream ks(fp);
    int last = 0;   // kseq's last_char
    for (;;) {
        int c;
        if (last == 0) {   // jump to the next header line
            while ((c = ks.getc()) >= 0 && c != '>' && c != '@') {}
            if (c < 0) break;
            last = c;
        }
        std::string hdr, name, seq;
        ks.rest_of_line(&hdr);
        size_t e = 0;
        while (e < hdr.size() && !isspace((unsigned char)hdr[e])) ++e;
        name = hdr.substr(0, e);
        while ((c = ks.getc()) >= 0 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq.push_back((char)c);
            ks.rest_of_line(&seq);
        }
        if (c == '>' || c == '@') last = c; else last = 0;
        if (c == '+') {   // FASTQ: skip the '+' line, then the quality block
            ks.rest_of_line(nullptr);
            std::string qual;
            while (qual.size() < seq.size()) {
                size_t before = qual.size();
                bool was_eof = ks.eof && ks.beg >= ks.end;
                ks.rest_of_line(&qual);
                if (was_eof && qual.size() == before) break;
                if (ks.eof && ks.beg >= ks.end && qual.size() == before) break;
            }
            last = 0;
        }
        out.emplace_back(std::move(name), std::move(seq));
        if (c < 0 && last == 0 && ks.eof && ks.beg >= ks.end) break;
    }
    gzclose(fp);
    return true;
}

}  // namespace dg
