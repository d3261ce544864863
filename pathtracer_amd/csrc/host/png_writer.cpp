// png_writer.cpp -- 8-bit RGBA PNG output: stands where the reference calls lodepng::encode (utils.cpp:3-14).
//
// Own, small encoder (nothing of lodepng is reused): per-row adaptive filtering (the filter type with the
// smallest sum of absolute residuals, the usual heuristic), LZ77 over a 32 KiB window with hash chains, and
// deflate blocks with dynamic Huffman codes (RFC 1951) inside a zlib stream (RFC 1950).  Pixels decode
// identically to what any other PNG writer would store; only the file size differs.
#include "png_writer.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

uint32_t crc_table[256];
bool crc_ready = false;

void crc_init() {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}

uint32_t crc32(const unsigned char* p, size_t n, uint32_t crc = 0xFFFFFFFFu) {
    if (!crc_ready) crc_init();
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

void put32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24));
    v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8));
    v.push_back((unsigned char)x);
}

void chunk(std::vector<unsigned char>& out, const char* type, const std::vector<unsigned char>& data) {
    put32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32(out.data() + start, out.size() - start) ^ 0xFFFFFFFFu);
}

// ---- scanline filters (PNG specification, section 9) --------------------------------------------
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

void filterRow(int type, const unsigned char* cur, const unsigned char* prev, size_t n, size_t bpp, unsigned char* out) {
    for (size_t i = 0; i < n; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        int pred = 0;
        switch (type) {
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: pred = paeth(a, b, c); break;
            default: break;
        }
        out[i] = (unsigned char)(cur[i] - pred);
    }
}

// ---- deflate ---------------------------------------------------------------------------------------
struct BitWriter {
    std::vector<unsigned char>& out;
    uint64_t acc = 0;
    int nbits = 0;
    explicit BitWriter(std::vector<unsigned char>& o) : out(o) {}
    void bits(uint32_t value, int count) {  // least significant bit first
        acc |= (uint64_t)value << nbits;
        nbits += count;
        while (nbits >= 8) {
            out.push_back((unsigned char)(acc & 0xFF));
            acc >>= 8;
            nbits -= 8;
        }
    }
    void code(uint32_t c, int len) {  // Huffman codes go most significant bit first
        uint32_t r = 0;
        for (int i = 0; i < len; ++i) r |= ((c >> i) & 1u) << (len - 1 - i);
        bits(r, len);
    }
    void flush() {
        if (nbits) bits(0, 8 - nbits);
    }
};

const int kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const int kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const int kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

int lenSymbol(int len) {
    int s = 28;
    while (kLenBase[s] > len) --s;
    return s;
}
int distSymbol(int dist) {
    int s = 29;
    while (kDistBase[s] > dist) --s;
    return s;
}

struct Token {
    uint16_t len;   // 0: literal
    uint16_t dist;  // literal: the byte
};

// Code lengths (at most `limit` bits) of a Huffman code for the given frequencies; symbols that do not occur get 0.
std::vector<int> codeLengths(const std::vector<uint32_t>& freq, int limit) {
    const int n = (int)freq.size();
    std::vector<int> len(n, 0);
    struct Node {
        uint64_t w;
        int left, right;
    };
    std::vector<Node> nodes;
    std::vector<int> alive;
    for (int i = 0; i < n; ++i)
        if (freq[i]) {
            nodes.push_back({freq[i], -1, i});
            alive.push_back((int)nodes.size() - 1);
        }
    if (alive.empty()) return len;
    if (alive.size() == 1) {
        len[nodes[alive[0]].right] = 1;
        return len;
    }
    auto heavier = [&](int a, int b) { return nodes[a].w > nodes[b].w || (nodes[a].w == nodes[b].w && a > b); };
    std::make_heap(alive.begin(), alive.end(), heavier);
    while (alive.size() > 1) {
        std::pop_heap(alive.begin(), alive.end(), heavier);
        const int a = alive.back();
        alive.pop_back();
        std::pop_heap(alive.begin(), alive.end(), heavier);
        const int b = alive.back();
        alive.pop_back();
        nodes.push_back({nodes[a].w + nodes[b].w, a, b});
        alive.push_back((int)nodes.size() - 1);
        std::push_heap(alive.begin(), alive.end(), heavier);
    }
    // depths
    std::vector<std::pair<int, int>> stack{{alive[0], 0}};
    while (!stack.empty()) {
        const auto [v, d] = stack.back();
        stack.pop_back();
        if (nodes[v].left < 0) len[nodes[v].right] = d;
        else {
            stack.push_back({nodes[v].left, d + 1});
            stack.push_back({nodes[v].right, d + 1});
        }
    }
    // limit the lengths: clamp, then restore the Kraft sum by lengthening the shortest-over-budget codes
    bool over = false;
    for (int i = 0; i < n; ++i)
        if (len[i] > limit) {
            len[i] = limit;
            over = true;
        }
    if (over) {
        auto kraft = [&]() {
            uint64_t k = 0;
            for (int i = 0; i < n; ++i)
                if (len[i]) k += 1ull << (limit - len[i]);
            return k;
        };
        const uint64_t full = 1ull << limit;
        while (kraft() > full) {  // lengthen the longest code shorter than the limit (cheapest in bits for rare symbols)
            int best = -1;
            for (int i = 0; i < n; ++i)
                if (len[i] && len[i] < limit && (best < 0 || len[i] > len[best] || (len[i] == len[best] && freq[i] < freq[best]))) best = i;
            if (best < 0) break;
            ++len[best];
        }
        // ... and make the code complete again (inflate rejects an under-subscribed set): shorten codes, largest
        // steps first, while they fit
        uint64_t deficit = full - kraft();
        for (int L = 2; L <= limit && deficit; ++L) {
            const uint64_t step = 1ull << (limit - L);  // what shortening a code of length L adds
            for (int i = 0; i < n && deficit >= step; ++i)
                if (len[i] == L) {
                    --len[i];
                    deficit -= step;
                }
        }
    }
    return len;
}

// Canonical codes for the lengths (RFC 1951, 3.2.2).
std::vector<uint32_t> canonicalCodes(const std::vector<int>& len) {
    int count[16] = {0};
    for (int l : len) count[l]++;
    count[0] = 0;
    uint32_t next[16] = {0}, code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + count[b - 1]) << 1;
        next[b] = code;
    }
    std::vector<uint32_t> codes(len.size(), 0);
    for (size_t i = 0; i < len.size(); ++i)
        if (len[i]) codes[i] = next[len[i]]++;
    return codes;
}

void writeBlock(BitWriter& bw, const Token* tok, size_t n, bool last) {
    std::vector<uint32_t> lf(286, 0), df(30, 0);
    for (size_t i = 0; i < n; ++i) {
        if (tok[i].len == 0) lf[tok[i].dist]++;
        else {
            lf[257 + lenSymbol(tok[i].len)]++;
            df[distSymbol(tok[i].dist)]++;
        }
    }
    lf[256] = 1;  // end of block
    std::vector<int> ll = codeLengths(lf, 15), dl = codeLengths(df, 15);
    bool any_dist = false;
    for (int l : dl) any_dist = any_dist || l;
    if (!any_dist) dl[0] = 1;  // at least one distance code has to be described
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
    // the code lengths, run-length coded with the symbols 16 (repeat previous 3-6), 17 (zeros 3-10), 18 (zeros 11-138)
    std::vector<int> seq(ll.begin(), ll.begin() + hlit);
    seq.insert(seq.end(), dl.begin(), dl.begin() + hdist);
    std::vector<std::pair<int, int>> rle;  // (symbol, extra value)
    for (size_t i = 0; i < seq.size();) {
        size_t run = 1;
        while (i + run < seq.size() && seq[i + run] == seq[i]) ++run;
        if (seq[i] == 0 && run >= 3) {
            const size_t r = std::min<size_t>(run, 138);
            rle.push_back(r <= 10 ? std::make_pair(17, (int)r - 3) : std::make_pair(18, (int)r - 11));
            i += r;
        } else if (run >= 4) {
            rle.push_back({seq[i], 0});
            const size_t r = std::min<size_t>(run - 1, 6);
            rle.push_back({16, (int)r - 3});
            i += 1 + r;
        } else {
            rle.push_back({seq[i], 0});
            ++i;
        }
    }
    std::vector<uint32_t> cf(19, 0);
    for (auto& p : rle) cf[p.first]++;
    std::vector<int> cl = codeLengths(cf, 7);
    static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[order[hclen - 1]] == 0) --hclen;
    const std::vector<uint32_t> lc = canonicalCodes(ll), dc = canonicalCodes(dl), cc = canonicalCodes(cl);

    bw.bits(last ? 1 : 0, 1);
    bw.bits(2, 2);  // dynamic Huffman codes
    bw.bits((uint32_t)(hlit - 257), 5);
    bw.bits((uint32_t)(hdist - 1), 5);
    bw.bits((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) bw.bits((uint32_t)cl[order[i]], 3);
    for (auto& p : rle) {
        bw.code(cc[p.first], cl[p.first]);
        if (p.first == 16) bw.bits((uint32_t)p.second, 2);
        else if (p.first == 17) bw.bits((uint32_t)p.second, 3);
        else if (p.first == 18) bw.bits((uint32_t)p.second, 7);
    }
    for (size_t i = 0; i < n; ++i) {
        if (tok[i].len == 0) {
            bw.code(lc[tok[i].dist], ll[tok[i].dist]);
        } else {
            const int ls = lenSymbol(tok[i].len), ds = distSymbol(tok[i].dist);
            bw.code(lc[257 + ls], ll[257 + ls]);
            if (kLenExtra[ls]) bw.bits((uint32_t)(tok[i].len - kLenBase[ls]), kLenExtra[ls]);
            bw.code(dc[ds], dl[ds]);
            if (kDistExtra[ds]) bw.bits((uint32_t)(tok[i].dist - kDistBase[ds]), kDistExtra[ds]);
        }
    }
    bw.code(lc[256], ll[256]);
}

// LZ77 with hash chains over a 32 KiB window, greedy, matches of 3..258 bytes.
std::vector<Token> lz77(const std::vector<unsigned char>& d) {
    const size_t n = d.size();
    std::vector<Token> tok;
    tok.reserve(n / 2 + 16);
    const int HASH_BITS = 15, WINDOW = 32768, MAX_CHAIN = 32;
    std::vector<int32_t> head(1u << HASH_BITS, -1), prev(n, -1);
    auto hash = [&](size_t i) { return ((uint32_t)(d[i] << 10) ^ (uint32_t)(d[i + 1] << 5) ^ d[i + 2]) & ((1u << HASH_BITS) - 1u); };
    size_t i = 0;
    while (i < n) {
        int best_len = 0, best_dist = 0;
        if (i + 3 <= n) {
            const uint32_t h = hash(i);
            int32_t j = head[h];
            int chain = 0;
            const size_t max_len = std::min<size_t>(258, n - i);
            while (j >= 0 && (int64_t)i - j <= WINDOW && chain++ < MAX_CHAIN) {
                size_t l = 0;
                while (l < max_len && d[j + l] == d[i + l]) ++l;
                if ((int)l > best_len) {
                    best_len = (int)l;
                    best_dist = (int)(i - j);
                    if (l == max_len) break;
                }
                j = prev[j];
            }
        }
        const size_t step = best_len >= 3 ? (size_t)best_len : 1;
        if (best_len >= 3) tok.push_back({(uint16_t)best_len, (uint16_t)best_dist});
        else tok.push_back({0, d[i]});
        for (size_t k = 0; k < step; ++k) {  // every position enters the chains
            if (i + k + 3 <= n) {
                const uint32_t h = hash(i + k);
                prev[i + k] = head[h];
                head[h] = (int32_t)(i + k);
            }
        }
        i += step;
    }
    return tok;
}

}  // namespace

bool png_write_rgba8(const std::string& filename, const unsigned char* rgba, uint32_t width, uint32_t height, std::string* error) {
    if (!rgba || width == 0 || height == 0) {
        if (error) *error = "empty image";
        return false;
    }
    // scanlines, each with the filter type that leaves the smallest residuals
    const size_t stride = (size_t)width * 4, bpp = 4;
    std::vector<unsigned char> raw;
    raw.reserve((size_t)height * (stride + 1));
    std::vector<unsigned char> cand(stride), best(stride);
    for (uint32_t y = 0; y < height; ++y) {
        const unsigned char* row = rgba + (size_t)y * stride;
        const unsigned char* prev = y ? row - stride : nullptr;
        int best_type = 0;
        uint64_t best_sum = ~0ull;
        for (int t = 0; t < 5; ++t) {
            filterRow(t, row, prev, stride, bpp, cand.data());
            uint64_t sum = 0;
            for (size_t i = 0; i < stride; ++i) sum += (uint64_t)std::abs((int)(signed char)cand[i]);
            if (sum < best_sum) {
                best_sum = sum;
                best_type = t;
                best.swap(cand);
            }
        }
        raw.push_back((unsigned char)best_type);
        raw.insert(raw.end(), best.begin(), best.end());
    }
    // zlib stream: header, deflate blocks of at most 65535 tokens with their own Huffman codes, adler32
    std::vector<unsigned char> z;
    z.push_back(0x78);
    z.push_back(0x9C);
    {
        const std::vector<Token> tok = lz77(raw);
        BitWriter bw(z);
        const size_t per_block = 65535;
        for (size_t at = 0; at < tok.size() || at == 0; at += per_block) {
            const size_t n = std::min(per_block, tok.size() - at);
            writeBlock(bw, tok.data() + at, n, at + n >= tok.size());
            if (tok.empty()) break;
        }
        bw.flush();
    }
    uint32_t a = 1, b = 0;  // adler32
    for (size_t i = 0; i < raw.size();) {
        const size_t n = std::min<size_t>(raw.size() - i, 5552);
        for (size_t k = 0; k < n; ++k) {
            a += raw[i + k];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        i += n;
    }
    put32(z, (b << 16) | a);

    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<unsigned char> ihdr;
    put32(ihdr, width);
    put32(ihdr, height);
    ihdr.push_back(8);  // bit depth
    ihdr.push_back(6);  // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});

    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) {
        if (error) *error = "cannot open " + filename + " for writing";
        return false;
    }
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok && error) *error = "short write to " + filename;
    return ok;
}
