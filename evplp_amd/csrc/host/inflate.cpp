// zlib (RFC 1950) / DEFLATE (RFC 1951) decompression for the PNG decoder.
#include "decoders.hpp"

#include <algorithm>
#include <stdexcept>

namespace evplp {
namespace {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t hold = 0; int nbits = 0;
    BitReader(const uint8_t *d, size_t n) : p(d), end(d + n) {}
    uint32_t bits(int n) {                       // LSB first
        while (nbits < n) {
            if (p >= end) throw std::runtime_error("deflate: input ends inside a block");
            hold |= (uint32_t)(*p++) << nbits; nbits += 8;
        }
        uint32_t v = n ? (hold & ((1u << n) - 1u)) : 0u;
        hold >>= n; nbits -= n;
        return v;
    }
    void align() { hold = 0; nbits = 0; }
};

// canonical Huffman code: symbols sorted by (length, value); decoded one bit at a time against the
// first code of each length
struct Huffman {
    uint16_t count[16] = { 0 };
    uint16_t symbol[288] = { 0 };
    void build(const uint8_t *lengths, int n) {
        for (int i = 0; i < 16; i++) count[i] = 0;
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left = (left << 1) - count[len];
            if (left < 0) throw std::runtime_error("deflate: over-subscribed Huffman code");
        }
        uint16_t offs[16]; offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; i++) if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            code |= (int)br.bits(1);
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        throw std::runtime_error("deflate: invalid Huffman code");
    }
};

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

void inflate_codes(BitReader &br, const Huffman &lit, const Huffman &dist, std::vector<uint8_t> &out) {
    for (;;) {
        int sym = lit.decode(br);
        if (sym < 256) { out.push_back((uint8_t)sym); continue; }
        if (sym == 256) return;
        sym -= 257;
        if (sym >= 29) throw std::runtime_error("deflate: bad length symbol");
        int len = kLenBase[sym] + (int)br.bits(kLenExtra[sym]);
        int ds = dist.decode(br);
        if (ds >= 30) throw std::runtime_error("deflate: bad distance symbol");
        size_t d = (size_t)kDistBase[ds] + br.bits(kDistExtra[ds]);
        if (d > out.size()) throw std::runtime_error("deflate: distance beyond the start of the output");
        size_t from = out.size() - d;
        for (int i = 0; i < len; i++) out.push_back(out[from + i]);   // may overlap: byte by byte
    }
}

} // namespace

std::vector<uint8_t> zlib_inflate(const uint8_t *data, size_t size, size_t size_hint) {
    if (size < 2) throw std::runtime_error("zlib: truncated header");
    int cmf = data[0], flg = data[1];
    if ((cmf & 15) != 8 || ((cmf << 8) | flg) % 31 != 0) throw std::runtime_error("zlib: bad header");
    if (flg & 32) throw std::runtime_error("zlib: preset dictionary not allowed");
    BitReader br(data + 2, size - 2);
    // (the hint comes from a header the file wrote itself: never more than a deflate stream of this size can hold, 1032 : 1)
    std::vector<uint8_t> out; out.reserve(std::min(size_hint, size * 1032 + 64));
    Huffman fixed_lit, fixed_dist; bool have_fixed = false;
    int last;
    do {
        last = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) throw std::runtime_error("deflate: truncated stored block");
            uint32_t len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((len ^ 0xffffu) != nlen) throw std::runtime_error("deflate: stored block length check failed");
            if ((size_t)(br.end - br.p) < len) throw std::runtime_error("deflate: truncated stored block");
            out.insert(out.end(), br.p, br.p + len); br.p += len;
        } else if (type == 1) {
            if (!have_fixed) {
                uint8_t l[288];
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                fixed_lit.build(l, 288);
                uint8_t d[30]; for (int i = 0; i < 30; i++) d[i] = 5;
                fixed_dist.build(d, 30);
                have_fixed = true;
            }
            inflate_codes(br, fixed_lit, fixed_dist, out);
        } else if (type == 2) {
            int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            if (nlen > 286 || ndist > 30) throw std::runtime_error("deflate: too many length / distance codes");
            static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
            uint8_t lengths[320] = { 0 };
            for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)br.bits(3);
            Huffman lencode; lencode.build(lengths, 19);
            uint8_t ll[320] = { 0 };
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = lencode.decode(br);
                if (sym < 16) ll[idx++] = (uint8_t)sym;
                else {
                    int prev = 0, rep;
                    if (sym == 16) { if (idx == 0) throw std::runtime_error("deflate: repeat without a previous length"); prev = ll[idx - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (idx + rep > nlen + ndist) throw std::runtime_error("deflate: too many code lengths");
                    while (rep--) ll[idx++] = (uint8_t)prev;
                }
            }
            if (ll[256] == 0) throw std::runtime_error("deflate: no end-of-block code");
            Huffman lit, dist; lit.build(ll, nlen); dist.build(ll + nlen, ndist);
            inflate_codes(br, lit, dist, out);
        } else throw std::runtime_error("deflate: reserved block type");
    } while (!last);
    return out;
}

} // namespace evplp
