#include "image_out.h"

#include <cmath>
#include <cstdio>

namespace trt {
namespace {

inline uint8_t encode(double x)
{
    // main.cpp:34: (unsigned char) clamp(pow(x, 1.0f/2.2f) * 255, 0.0, 255.0)
    double v = std::pow(x, (double)(1.0f / 2.2f)) * 255;
    if (!(v > 0.0)) v = 0.0;   // also maps NaN to 0
    if (v > 255.0) v = 255.0;
    return (uint8_t)v;
}

struct Crc {
    uint32_t table[256];
    Crc()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
    }
};

struct ChunkWriter {
    FILE* fp;
    uint32_t crc = 0xFFFFFFFFu;
    static const Crc& tab() { static Crc c; return c; }
    void raw(const uint8_t* p, size_t n) { std::fwrite(p, 1, n, fp); }
    void put(const uint8_t* p, size_t n)
    {
        raw(p, n);
        for (size_t i = 0; i < n; ++i) crc = tab().table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    }
    void u32raw(uint32_t v) { uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; raw(b, 4); }
    void u32(uint32_t v) { uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; put(b, 4); }
    void begin(const char* type, uint32_t len) { u32raw(len); crc = 0xFFFFFFFFu; put((const uint8_t*)type, 4); }
    void end() { u32raw(~crc); }
};

}  // namespace

void tonemap(const double* s, int w, int h, std::vector<uint8_t>& out)
{
    out.resize((size_t)w * h * 3);
    for (size_t i = 0; i < out.size(); ++i) out[i] = encode(s[i]);
}

void tonemap(const float* s, int w, int h, std::vector<uint8_t>& out)
{
    out.resize((size_t)w * h * 3);
    for (size_t i = 0; i < out.size(); ++i) out[i] = encode((double)s[i]);
}

bool writePNG(const std::string& path, int w, int h, const uint8_t* rgb)
{
    if (w <= 0 || h <= 0) return false;
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    ChunkWriter cw{fp};
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    cw.raw(sig, 8);
    cw.begin("IHDR", 13);
    cw.u32((uint32_t)w);
    cw.u32((uint32_t)h);
    const uint8_t hdr[5] = {8, 2, 0, 0, 0};  // 8-bit, truecolour, deflate, no filter method, no interlace
    cw.put(hdr, 5);
    cw.end();

    // zlib stream of stored blocks over the filtered scanlines (filter byte 0 + 3w bytes each)
    const size_t pitch = (size_t)w * 3 + 1;
    const size_t raw_len = pitch * (size_t)h;
    const size_t n_blocks = (raw_len + 65534) / 65535;
    const size_t idat_len = 2 + n_blocks * 5 + raw_len + 4;
    if (idat_len > 0x7FFFFFFFu) { std::fclose(fp); return false; }
    cw.begin("IDAT", (uint32_t)idat_len);
    const uint8_t zh[2] = {0x78, 0x01};
    cw.put(zh, 2);
    uint32_t a = 1, b = 0;  // adler32
    std::vector<uint8_t> line(pitch);
    size_t in_block = 0, remaining = raw_len;
    for (int y = 0; y < h; ++y) {
        line[0] = 0;
        const uint8_t* src = rgb + (size_t)y * w * 3;
        for (size_t i = 0; i < (size_t)w * 3; ++i) line[i + 1] = src[i];
        size_t off = 0;
        while (off < pitch) {
            if (in_block == 0) {
                const size_t len = remaining < 65535 ? remaining : 65535;
                const uint8_t bh[5] = {(uint8_t)(remaining <= 65535 ? 1 : 0), (uint8_t)(len & 0xFF), (uint8_t)(len >> 8), (uint8_t)(~len & 0xFF), (uint8_t)((~len >> 8) & 0xFF)};
                cw.put(bh, 5);
                in_block = len;
            }
            const size_t take = (pitch - off) < in_block ? (pitch - off) : in_block;
            cw.put(line.data() + off, take);
            for (size_t i = 0; i < take; ++i) { a = (a + line[off + i]) % 65521u; b = (b + a) % 65521u; }
            off += take;
            in_block -= take;
            remaining -= take;
        }
    }
    cw.u32((b << 16) | a);
    cw.end();
    cw.begin("IEND", 0);
    cw.end();
    const bool ok = std::ferror(fp) == 0;
    std::fclose(fp);
    return ok;
}

bool imshow(const double* src, const std::string& basedir, const std::string& index, int w, int h)
{
    std::vector<uint8_t> bytes;
    tonemap(src, w, h, bytes);
    const std::string name = basedir + "/image" + index + ".png";
    const bool ok = writePNG(name, w, h, bytes.data());
    if (ok) std::printf("\nImage output to %s\n", name.c_str());
    return ok;
}

}  // namespace trt
