// png.cpp — PNG decoder for map_Kd textures.
//
// The reference loads textures with cv::imread(path) (material.cpp:6), whose default flag IMREAD_COLOR always yields 8-bit, 3 channels.
// For a PNG that is libpng with these transforms (OpenCV's PngDecoder): palette -> RGB, grey of 1 / 2 / 4 bits expanded to 8
// (bit replication: x * 255 / (2^depth - 1)), 16-bit samples stripped to their high byte, alpha (and tRNS) DROPPED, not blended, grey
// replicated into three channels; no gamma.  This file restates exactly that, dependency-free (zlib / libpng headers are not part of
// the build): inflate per RFC 1951 (stored, fixed and dynamic Huffman blocks), the five scanline filters, Adam7 interlace.
// tests/test_loaders.py checks it against PIL on every colour type and depth, interlaced files and corrupt input.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace trt {
namespace {

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int have = 0;
    bool bad = false;
    uint32_t bits(int k)
    {
        while (have < k) {
            if (pos >= n) { bad = true; return 0; }
            acc |= (uint32_t)p[pos++] << have;
            have += 8;
        }
        const uint32_t v = k ? (acc & ((1u << k) - 1u)) : 0u;
        acc >>= k;
        have -= k;
        return v;
    }
    void alignByte() { acc = 0; have = 0; }
};

// canonical Huffman code over `count` symbols with the given lengths (<= 15 bits); decode() walks it bit by bit
struct Huffman {
    uint16_t counts[16] = {0}, symbols[288] = {0};
    bool build(const uint8_t* lengths, int count)
    {
        std::memset(counts, 0, sizeof(counts));
        for (int i = 0; i < count; ++i) counts[lengths[i]]++;
        counts[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) {
            left <<= 1;
            left -= counts[l];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + counts[l];
        for (int i = 0; i < count; ++i)
            if (lengths[i]) symbols[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= (int)br.bits(1);
            if (br.bad) return -1;
            const int c = counts[l];
            if (code - c < first) return symbols[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect)
{
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n < 6) return false;
    if ((src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;  // zlib header: deflate, no preset dictionary
    BitReader br{src + 2, n - 2};
    out.clear();
    out.reserve(expect);
    for (;;) {
        const uint32_t last = br.bits(1), type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.alignByte();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > br.n || out.size() + len > expect) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[ORDER[i]] = (uint8_t)br.bits(3);
                Huffman lencode;
                if (br.bad || !lencode.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                    int rep, val = 0;
                    if (sym == 16) { if (idx == 0) return false; val = lengths[idx - 1]; rep = 3 + (int)br.bits(2); }
                    else if (sym == 17) rep = 3 + (int)br.bits(3);
                    else rep = 11 + (int)br.bits(7);
                    if (br.bad || idx + rep > nlen + ndist) return false;
                    while (rep--) lengths[idx++] = (uint8_t)val;
                }
                if (lengths[256] == 0 || !lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) {
                    if (out.size() >= expect) return false;
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    if (sym > 285) return false;
                    const size_t len = LBASE[sym - 257] + br.bits(LEXT[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const size_t d = DBASE[ds] + br.bits(DEXT[ds]);
                    if (br.bad || d > out.size() || out.size() + len > expect) return false;
                    const size_t from = out.size() - d;
                    for (size_t i = 0; i < len; ++i) out.push_back(out[from + i]);  // (byte by byte: the ranges may overlap)
                }
            }
        } else {
            return false;
        }
        if (last) break;
    }
    return out.size() == expect;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// un-filters `rows` scanlines of `stride` bytes each (every one preceded by its filter byte) in place; bpp = bytes per complete pixel, at least 1
bool unfilter(uint8_t* data, size_t rows, size_t stride, int bpp)
{
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t* prev = zero.data();
    for (size_t y = 0; y < rows; ++y) {
        uint8_t* line = data + y * (stride + 1);
        const int ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= (size_t)bpp ? prev[x - bpp] : 0;
            int add;
            switch (ft) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = b; break;
                case 3: add = (a + b) >> 1; break;
                case 4: add = paeth(a, b, c); break;
                default: return false;
            }
            cur[x] = (uint8_t)(cur[x] + add);
        }
        prev = cur;
    }
    return true;
}

}  // namespace

// Decodes a PNG file into 8-bit RGB, row 0 = first row of the file, the way cv::imread(path) (IMREAD_COLOR) presents it (header).
bool decodePNG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height)
{
    std::FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> file;
    {
        uint8_t buf[65536];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof(buf), fp)) > 0) {
            file.insert(file.end(), buf, buf + got);
            if (file.size() > (1u << 30)) break;
        }
        std::fclose(fp);
    }
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), SIG, 8) != 0) return false;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_ihdr = false, seen_end = false;
    for (size_t at = 8; at + 12 <= file.size() && !seen_end;) {
        const uint32_t len = be32(&file[at]);
        if (len > file.size() - at - 12) return false;
        const uint8_t* type = &file[at + 4];
        const uint8_t* data = &file[at + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13 || seen_ihdr) return false;
            w = be32(data); h = be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0 || interlace > 1) return false;
            seen_ihdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        at += 12 + (size_t)len;
    }
    if (!seen_ihdr || idat.empty() || w == 0 || h == 0 || w > 32768 || h > 32768) return false;
    int channels;
    switch (ctype) {
        case 0: channels = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16) return false; break;
        case 2: channels = 3; if (depth != 8 && depth != 16) return false; break;
        case 3: channels = 1; if (depth != 1 && depth != 2 && depth != 4 && depth != 8) return false; if (plte.size() < 3 || plte.size() % 3) return false; break;
        case 4: channels = 2; if (depth != 8 && depth != 16) return false; break;
        case 6: channels = 4; if (depth != 8 && depth != 16) return false; break;
        default: return false;
    }
    const int bits_pp = channels * depth, bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // the passes: the whole image, or Adam7's seven sub-images
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass ADAM7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass WHOLE = {0, 0, 1, 1};
    const int n_pass = interlace ? 7 : 1;
    size_t total = 0;
    size_t pw[7], ph[7], stride[7];
    for (int k = 0; k < n_pass; ++k) {
        const Pass& P = interlace ? ADAM7[k] : WHOLE;
        pw[k] = w > P.x0 ? (w - P.x0 + P.dx - 1) / P.dx : 0;
        ph[k] = h > P.y0 ? (h - P.y0 + P.dy - 1) / P.dy : 0;
        stride[k] = (pw[k] * (size_t)bits_pp + 7) / 8;
        if (pw[k] && ph[k]) total += ph[k] * (stride[k] + 1);
    }
    // deflate expands by less than 1032 : 1 (a 258-byte match per 2 bits): a stream too short for the scanlines the header announces is rejected before
    // anything is allocated for them.  (+ a cap of 2^28 pixels.)
    if ((uint64_t)w * h > (1ull << 28) || total / 1032 > idat.size()) return false;
    std::vector<uint8_t> raw;
    if (!inflate(idat.data(), idat.size(), raw, total)) return false;
    rgb.assign((size_t)w * h * 3, 0);
    const int maxv = (1 << (depth < 8 ? depth : 8)) - 1;
    size_t off = 0;
    for (int k = 0; k < n_pass; ++k) {
        if (!pw[k] || !ph[k]) continue;
        const Pass& P = interlace ? ADAM7[k] : WHOLE;
        uint8_t* base = raw.data() + off;
        if (!unfilter(base, ph[k], stride[k], bpp)) return false;
        for (size_t y = 0; y < ph[k]; ++y) {
            const uint8_t* line = base + y * (stride[k] + 1) + 1;
            for (size_t x = 0; x < pw[k]; ++x) {
                // the samples of this pixel as 8-bit values: high byte of a 16-bit sample; a sample of 1 / 2 / 4 bits as it is (scaled below)
                uint8_t s[4] = {0, 0, 0, 0};
                if (depth == 16) { for (int c = 0; c < channels; ++c) s[c] = line[(x * channels + c) * 2]; }
                else if (depth == 8) { for (int c = 0; c < channels; ++c) s[c] = line[x * channels + c]; }
                else {
                    const size_t bit = x * (size_t)depth;
                    s[0] = (uint8_t)((line[bit >> 3] >> (8 - depth - (bit & 7))) & maxv);
                }
                uint8_t* px = &rgb[(((size_t)P.y0 + y * P.dy) * w + (P.x0 + x * P.dx)) * 3];
                if (ctype == 3) {
                    const size_t i = (size_t)s[0] * 3;
                    if (i + 2 < plte.size()) { px[0] = plte[i]; px[1] = plte[i + 1]; px[2] = plte[i + 2]; }  // (an index past the palette: black, as libpng leaves it)
                } else if (ctype == 0 || ctype == 4) {
                    const uint8_t g = depth < 8 ? (uint8_t)(s[0] * 255 / maxv) : s[0];
                    px[0] = px[1] = px[2] = g;
                } else {
                    px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
                }
            }
        }
        off += ph[k] * (stride[k] + 1);
    }
    width = (int)w;
    height = (int)h;
    return true;
}

}  // namespace trt
