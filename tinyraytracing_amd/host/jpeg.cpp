// jpeg.cpp — JPEG decoder (baseline and progressive Huffman, 8-bit, grey or YCbCr) for map_Kd textures.
//
// The reference loads textures with cv::imread (material.cpp:6), i.e. libjpeg's default decode.  OpenCV and
// libjpeg headers are not available to this build, so the decode is restated here with libjpeg's own
// arithmetic, which makes the texels bit-identical to what the reference sees:
//   * Huffman / dequantisation per ITU T.81 (8-bit; SOF0 baseline in one interleaved scan, SOF2 progressive in any sequence of scans; restart intervals),
//   * the "islow" integer inverse DCT (13-bit constants, 2 extra bits kept between the passes),
//   * "fancy" (triangle-filter) chroma upsampling for 2x horizontal and/or 2x vertical subsampling,
//   * YCbCr -> RGB with 16-bit fixed-point tables.
// tests/test_loaders.py checks the output against PIL's (libjpeg-turbo's) decode of the shipped textures.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace trt {
namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int width = 0, height = 0;    // true downsampled size
    int stride = 0, rows = 0;     // allocated (multiple of 8 * blocks per MCU)
    std::vector<uint8_t> plane;   // decoded samples
    int pred = 0;
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill()
    {
        while (nbits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    const int b2 = p < end ? *p : 0;
                    if (b2 == 0) ++p;                 // stuffed zero
                    else { hit_marker = true; --p; byte = 0; }  // a marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)byte << (24 - nbits);
            nbits += 8;
        }
    }
    int get(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

int decodeSymbol(BitReader& br, const Huff& h)
{
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.get(1);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clampSample(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// 32-bit two's-complement arithmetic that WRAPS: what libjpeg's INT32 code does in every build of it on out-of-range coefficients (a corrupt or hostile
// file can hold any 16-bit value times any quantiser), spelled so that it is defined behaviour here.  Identical to plain int32_t wherever nothing overflows,
// i.e. on every valid file.
struct w32 {
    uint32_t v;
    w32() = default;
    w32(int32_t x) : v((uint32_t)x) {}
    static w32 raw(uint32_t u) { w32 r; r.v = u; return r; }
    friend w32 operator+(w32 a, w32 b) { return raw(a.v + b.v); }
    friend w32 operator-(w32 a, w32 b) { return raw(a.v - b.v); }
    friend w32 operator*(w32 a, w32 b) { return raw(a.v * b.v); }
    w32& operator+=(w32 b) { v += b.v; return *this; }
    w32& operator*=(w32 b) { v *= b.v; return *this; }
};
inline int32_t descale(w32 x, int n) { return (int32_t)(x.v + (1u << (n - 1))) >> n; }

// libjpeg jidctint.c (JDCT_ISLOW): Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13, PASS1_BITS = 2.
void idctIslow(const int32_t* coef, uint8_t* out, int stride)
{
    constexpr int CB = 13, P1 = 2;
    constexpr int32_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                      F2053 = 16819, F2562 = 20995, F3072 = 25172;
    w32 ws[64];
    for (int c = 0; c < 8; ++c) {
        const int32_t* in = coef + c;
        w32 z2 = w32(in[16]), z3 = w32(in[48]);
        w32 z1 = (z2 + z3) * w32(F0541);
        w32 tmp2 = z1 + z3 * w32(-F1847);
        w32 tmp3 = z1 + z2 * w32(F0765);
        z2 = w32(in[0]);
        z3 = w32(in[32]);
        w32 tmp0 = (z2 + z3) * w32(1 << CB), tmp1 = (z2 - z3) * w32(1 << CB);
        const w32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w32(in[56]); tmp1 = w32(in[40]); tmp2 = w32(in[24]); tmp3 = w32(in[8]);
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        w32 z4 = tmp1 + tmp3;
        const w32 z5 = (z3 + z4) * w32(F1175);
        tmp0 *= w32(F0298); tmp1 *= w32(F2053); tmp2 *= w32(F3072); tmp3 *= w32(F1501);
        z1 *= w32(-F0899); z2 *= w32(-F2562); z3 *= w32(-F1961); z4 *= w32(-F0390);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = w32(descale(tmp10 + tmp3, CB - P1));       ws[56 + c] = w32(descale(tmp10 - tmp3, CB - P1));
        ws[8 + c] = w32(descale(tmp11 + tmp2, CB - P1));   ws[48 + c] = w32(descale(tmp11 - tmp2, CB - P1));
        ws[16 + c] = w32(descale(tmp12 + tmp1, CB - P1));  ws[40 + c] = w32(descale(tmp12 - tmp1, CB - P1));
        ws[24 + c] = w32(descale(tmp13 + tmp0, CB - P1));  ws[32 + c] = w32(descale(tmp13 - tmp0, CB - P1));
    }
    for (int r = 0; r < 8; ++r) {
        const w32* w = ws + r * 8;
        uint8_t* o = out + r * stride;
        w32 z2 = w[2], z3 = w[6];
        w32 z1 = (z2 + z3) * w32(F0541);
        w32 tmp2 = z1 + z3 * w32(-F1847);
        w32 tmp3 = z1 + z2 * w32(F0765);
        w32 tmp0 = (w[0] + w[4]) * w32(1 << CB), tmp1 = (w[0] - w[4]) * w32(1 << CB);
        const w32 tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        w32 z4 = tmp1 + tmp3;
        const w32 z5 = (z3 + z4) * w32(F1175);
        tmp0 *= w32(F0298); tmp1 *= w32(F2053); tmp2 *= w32(F3072); tmp3 *= w32(F1501);
        z1 *= w32(-F0899); z2 *= w32(-F2562); z3 *= w32(-F1961); z4 *= w32(-F0390);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        o[0] = clampSample(descale(tmp10 + tmp3, S) + 128); o[7] = clampSample(descale(tmp10 - tmp3, S) + 128);
        o[1] = clampSample(descale(tmp11 + tmp2, S) + 128); o[6] = clampSample(descale(tmp11 - tmp2, S) + 128);
        o[2] = clampSample(descale(tmp12 + tmp1, S) + 128); o[5] = clampSample(descale(tmp12 - tmp1, S) + 128);
        o[3] = clampSample(descale(tmp13 + tmp0, S) + 128); o[4] = clampSample(descale(tmp13 - tmp0, S) + 128);
    }
}

// libjpeg jdsample.c "fancy" upsampling.  `in` has true size cw x ch; result is (cw*hs) x (ch*vs), cropped by the caller.
void upsampleFancy(const Component& c, int hs, int vs, std::vector<uint8_t>& out, int& ow, int& oh)
{
    const int cw = c.width, ch = c.height;
    ow = cw * hs;
    oh = ch * vs;
    out.resize((size_t)ow * oh);
    auto row = [&](int y) { return c.plane.data() + (size_t)(y < 0 ? 0 : (y >= ch ? ch - 1 : y)) * c.stride; };
    if (hs == 1 && vs == 1) {
        for (int y = 0; y < ch; ++y) std::memcpy(&out[(size_t)y * ow], row(y), (size_t)cw);
        return;
    }
    if (hs == 2 && vs == 1) {  // h2v1_fancy_upsample
        for (int y = 0; y < ch; ++y) {
            const uint8_t* in = row(y);
            uint8_t* o = &out[(size_t)y * ow];
            if (cw == 1) { o[0] = o[1] = in[0]; continue; }
            o[0] = in[0];
            o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
            for (int x = 1; x < cw - 1; ++x) {
                const int v = in[x] * 3;
                o[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
                o[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
            }
            o[2 * (cw - 1)] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
            o[2 * (cw - 1) + 1] = in[cw - 1];
        }
        return;
    }
    if (hs == 1 && vs == 2) {  // h1v2_fancy_upsample
        for (int y = 0; y < ch; ++y)
            for (int v = 0; v < 2; ++v) {
                const uint8_t* in0 = row(y);
                const uint8_t* in1 = row(v == 0 ? y - 1 : y + 1);
                uint8_t* o = &out[(size_t)(2 * y + v) * ow];
                const int bias = v == 0 ? 1 : 2;
                for (int x = 0; x < cw; ++x) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
            }
        return;
    }
    // h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16 triangle filter
    for (int y = 0; y < ch; ++y)
        for (int v = 0; v < 2; ++v) {
            const uint8_t* in0 = row(y);
            const uint8_t* in1 = row(v == 0 ? y - 1 : y + 1);
            uint8_t* o = &out[(size_t)(2 * y + v) * ow];
            if (cw == 1) {
                const int s = in0[0] * 3 + in1[0];
                o[0] = (uint8_t)((s * 4 + 8) >> 4);
                o[1] = (uint8_t)((s * 4 + 7) >> 4);
                continue;
            }
            int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
            o[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
            o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
            lastcol = thiscol;
            thiscol = nextcol;
            for (int x = 1; x < cw - 1; ++x) {
                nextcol = in0[x + 1] * 3 + in1[x + 1];
                o[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                o[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                lastcol = thiscol;
                thiscol = nextcol;
            }
            o[2 * (cw - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
            o[2 * (cw - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        }
}

// planes of samples -> interleaved RGB: grey replicated; else fancy upsampling of the chroma planes and libjpeg's YCbCr -> RGB tables
void finishImage(std::vector<Component>& comps, int hmax, int vmax, int width, int height, std::vector<uint8_t>& rgb)
{
    rgb.resize((size_t)width * height * 3);
    if (comps.size() == 1) {
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                const uint8_t g = comps[0].plane[(size_t)y * comps[0].stride + x];
                uint8_t* o = &rgb[((size_t)y * width + x) * 3];
                o[0] = o[1] = o[2] = g;
            }
        return;
    }
    std::vector<uint8_t> up[3];
    int uw[3], uh[3];
    for (int k = 0; k < 3; ++k) upsampleFancy(comps[k], hmax / comps[k].h, vmax / comps[k].v, up[k], uw[k], uh[k]);
    // jdcolor.c ycc_rgb_convert: 16-bit fixed point tables
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        const int x = i - 128;
        cr_r[i] = (int)((91881 * x + 32768) >> 16);    // FIX(1.40200)
        cb_b[i] = (int)((116130 * x + 32768) >> 16);   // FIX(1.77200)
        cr_g[i] = -46802 * x;                           // FIX(0.71414)
        cb_g[i] = -22554 * x + 32768;                   // FIX(0.34414)
    }
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            const int Y = up[0][(size_t)y * uw[0] + x], cb = up[1][(size_t)y * uw[1] + x], cr = up[2][(size_t)y * uw[2] + x];
            uint8_t* o = &rgb[((size_t)y * width + x) * 3];
            o[0] = clampSample(Y + cr_r[cr]);
            o[1] = clampSample(Y + ((cb_g[cb] + cr_g[cr]) >> 16));
            o[2] = clampSample(Y + cb_b[cb]);
        }
}

// ---- progressive JPEG (SOF2; ITU T.81 annex G, decoded the way libjpeg's jdphuff.c does): the scans fill a coefficient array per component — DC first / refine,
// AC first (with end-of-band runs) / refine —, then every block is dequantised and goes through the same IDCT, upsampling and colour conversion as a baseline
// file, so the texels are libjpeg's here too.
struct ProgComp {
    int bw = 0, bh = 0;            // blocks allocated (MCU-padded)
    std::vector<int16_t> coef;     // bw * bh * 64, natural (not zigzag) order
};

bool decodeProgressive(const std::vector<uint8_t>& data, size_t pos, uint16_t qt[4][64], Huff dc[4], Huff ac[4], std::vector<Component>& comps, int restart_interval,
                       int width, int height, std::vector<uint8_t>& rgb)
{
    int hmax = 1, vmax = 1;
    for (auto& c : comps) {
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) return false;
        hmax = c.h > hmax ? c.h : hmax;
        vmax = c.v > vmax ? c.v : vmax;
    }
    for (size_t k = 1; k < comps.size(); ++k)
        if (comps[k].h != 1 || comps[k].v != 1) return false;
    if (comps.size() == 1) { hmax = comps[0].h = 1; vmax = comps[0].v = 1; }
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    std::vector<ProgComp> pc(comps.size());
    for (size_t k = 0; k < comps.size(); ++k) {
        Component& c = comps[k];
        c.width = (width * c.h + hmax - 1) / hmax;
        c.height = (height * c.v + vmax - 1) / vmax;
        c.stride = mcux * c.h * 8;
        c.rows = mcuy * c.v * 8;
        pc[k].bw = mcux * c.h;
        pc[k].bh = mcuy * c.v;
        pc[k].coef.assign((size_t)pc[k].bw * pc[k].bh * 64, 0);
    }
    bool any_scan = false;
    while (pos + 4 <= data.size()) {
        if (data[pos] != 0xFF) return false;
        const int marker = data[pos + 1];
        if (marker == 0xFF) { ++pos; continue; }
        if (marker == 0xD9) break;  // EOI
        const size_t len = ((size_t)data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + len > data.size()) return false;
        const uint8_t* seg = &data[pos + 4];
        const size_t seglen = len - 2;
        if (marker == 0xDB) {
            size_t i = 0;
            while (i < seglen) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > seglen) return false;
                for (int k = 0; k < 64; ++k) { qt[tq][kZigzag[k]] = pq ? (uint16_t)((seg[i] << 8) | seg[i + 1]) : seg[i]; i += pq ? 2 : 1; }
            }
        } else if (marker == 0xC4) {
            size_t i = 0;
            while (i + 17 <= seglen) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                if (th > 3 || tc > 1) return false;
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[i + l]; total += h.bits[l]; }
                i += 17;
                if (total > 256 || i + (size_t)total > seglen) return false;
                std::memcpy(h.vals, seg + i, (size_t)total);
                i += (size_t)total;
                h.build();
                h.present = true;
            }
        } else if (marker == 0xDD) {
            if (seglen < 2) return false;
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {
            if (seglen < 1) return false;
            const int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size() || seglen < (size_t)(1 + 2 * ns + 3)) return false;
            int idx[4];
            for (int k = 0; k < ns; ++k) {
                idx[k] = -1;
                for (size_t c = 0; c < comps.size(); ++c) if (comps[c].id == seg[1 + 2 * k]) idx[k] = (int)c;
                if (idx[k] < 0) return false;
                comps[idx[k]].td = seg[2 + 2 * k] >> 4;
                comps[idx[k]].ta = seg[2 + 2 * k] & 15;
                if (comps[idx[k]].td > 3 || comps[idx[k]].ta > 3) return false;
            }
            const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
            if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) return false;
            for (int k = 0; k < ns; ++k) {
                if (Ss == 0 && Ah == 0 && !dc[comps[idx[k]].td].present) return false;
                if (Ss > 0 && !ac[comps[idx[k]].ta].present) return false;
            }
            const uint8_t* scan = &data[pos + 2 + len];
            BitReader br{scan, data.data() + data.size()};
            for (auto& c : comps) c.pred = 0;
            int eobrun = 0, restarts_left = restart_interval;
            const bool interleaved = ns > 1;
            // the scan's units: MCUs (interleaved) or the blocks that cover the component's true size (one component)
            const Component& c0 = comps[idx[0]];
            const int ux = interleaved ? mcux : (c0.width + 7) / 8, uy = interleaved ? mcuy : (c0.height + 7) / 8;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            auto refineNonZero = [&](int16_t& cf) {
                if (br.get(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
            };
            for (int yy = 0; yy < uy; ++yy)
                for (int xx = 0; xx < ux; ++xx) {
                    if (restart_interval && restarts_left == 0) {
                        const uint8_t* q = br.p;  // (the reader never runs past a marker: it stops in front of it and feeds zeros)
                        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                        if (q + 1 >= br.end) return false;
                        br.p = q + 2;
                        br.reset();
                        for (auto& c : comps) c.pred = 0;
                        eobrun = 0;
                        restarts_left = restart_interval;
                    }
                    for (int k = 0; k < ns; ++k) {
                        Component& c = comps[idx[k]];
                        ProgComp& P = pc[idx[k]];
                        const int nbx = interleaved ? c.h : 1, nby = interleaved ? c.v : 1;
                        for (int by = 0; by < nby; ++by)
                            for (int bx = 0; bx < nbx; ++bx) {
                                const int bxx = interleaved ? xx * c.h + bx : xx, byy = interleaved ? yy * c.v + by : yy;
                                int16_t* blk = &P.coef[((size_t)byy * P.bw + bxx) * 64];
                                if (Ss == 0) {
                                    if (Ah == 0) {
                                        const int t = decodeSymbol(br, dc[c.td]);
                                        if (t < 0 || t > 15) return false;
                                        c.pred += t ? extend(br.get(t), t) : 0;
                                        blk[0] = (int16_t)(c.pred * (1 << Al));
                                    } else if (br.get(1)) {
                                        blk[0] = (int16_t)(blk[0] | p1);
                                    }
                                } else if (Ah == 0) {
                                    if (eobrun > 0) { --eobrun; continue; }
                                    for (int kk = Ss; kk <= Se;) {
                                        const int rs = decodeSymbol(br, ac[c.ta]);
                                        if (rs < 0) return false;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.get(r); break; }
                                            kk += 16;
                                        } else {
                                            kk += r;
                                            if (kk > 63) return false;
                                            blk[kZigzag[kk]] = (int16_t)(extend(br.get(sz), sz) * (1 << Al));
                                            ++kk;
                                        }
                                    }
                                } else {
                                    int kk = Ss;
                                    if (eobrun == 0) {
                                        for (; kk <= Se; ++kk) {
                                            const int rs = decodeSymbol(br, ac[c.ta]);
                                            if (rs < 0) return false;
                                            int r = rs >> 4, sz = rs & 15;
                                            if (sz) {
                                                sz = br.get(1) ? p1 : m1;  // (a valid stream has size 1 here)
                                            } else if (r != 15) {
                                                eobrun = 1 << r;
                                                if (r) eobrun += br.get(r);
                                                break;
                                            }
                                            do {
                                                int16_t& cf = blk[kZigzag[kk]];
                                                if (cf != 0) refineNonZero(cf);
                                                else if (--r < 0) break;
                                                ++kk;
                                            } while (kk <= Se);
                                            if (sz && kk <= 63) blk[kZigzag[kk]] = (int16_t)sz;
                                        }
                                    }
                                    if (eobrun > 0) {
                                        for (; kk <= Se; ++kk) {
                                            int16_t& cf = blk[kZigzag[kk]];
                                            if (cf != 0) refineNonZero(cf);
                                        }
                                        --eobrun;
                                    }
                                }
                            }
                    }
                    if (restart_interval) --restarts_left;
                }
            any_scan = true;
            // on to the next marker behind the entropy-coded data (0xFF00 is a stuffed byte, 0xFFD0-D7 restart markers belong to the scan)
            size_t q = pos + 2 + len;
            while (q + 1 < data.size() && !(data[q] == 0xFF && data[q + 1] != 0x00 && !(data[q + 1] >= 0xD0 && data[q + 1] <= 0xD7) && data[q + 1] != 0xFF)) ++q;
            pos = q;
            continue;
        }
        pos += 2 + len;
    }
    if (!any_scan) return false;
    int32_t block[64];
    for (size_t k = 0; k < comps.size(); ++k) {
        Component& c = comps[k];
        c.plane.assign((size_t)c.stride * c.rows, 0);
        for (int by = 0; by < pc[k].bh; ++by)
            for (int bx = 0; bx < pc[k].bw; ++bx) {
                const int16_t* blk = &pc[k].coef[((size_t)by * pc[k].bw + bx) * 64];
                for (int i = 0; i < 64; ++i) block[i] = (int32_t)blk[i] * qt[c.tq][i];
                idctIslow(block, c.plane.data() + (size_t)(by * 8) * c.stride + (size_t)bx * 8, c.stride);
            }
    }
    finishImage(comps, hmax, vmax, width, height, rgb);
    return true;
}

}  // namespace

bool decodeJPEG(const std::string& path, std::vector<uint8_t>& rgb, int& width, int& height)
{
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> data;
    {
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), fp)) > 0) data.insert(data.end(), buf, buf + n);
    }
    std::fclose(fp);
    if (data.size() < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;

    uint16_t qt[4][64] = {};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int restart_interval = 0;
    size_t pos = 2;
    const uint8_t* scan = nullptr;
    bool progressive = false;
    while (pos + 4 <= data.size()) {
        if (data[pos] != 0xFF) return false;
        const int marker = data[pos + 1];
        if (marker == 0xFF) { ++pos; continue; }
        const size_t len = ((size_t)data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + len > data.size()) return false;
        const uint8_t* seg = &data[pos + 4];
        const size_t seglen = len - 2;
        if (marker == 0xDB) {
            size_t i = 0;
            while (i < seglen) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > seglen) return false;
                for (int k = 0; k < 64; ++k) {
                    qt[tq][kZigzag[k]] = pq ? (uint16_t)((seg[i] << 8) | seg[i + 1]) : seg[i];
                    i += pq ? 2 : 1;
                }
            }
        } else if (marker == 0xC0 || marker == 0xC2) {
            progressive = marker == 0xC2;
            if (seglen < 6 || seg[0] != 8) return false;
            height = (seg[1] << 8) | seg[2];
            width = (seg[3] << 8) | seg[4];
            const int nc = seg[5];
            if ((nc != 1 && nc != 3) || seglen < (size_t)(6 + 3 * nc) || width <= 0 || height <= 0) return false;
            // A header may claim any size up to 65535 x 65535; every 8x8 block of it costs at least one bit of entropy-coded data (its DC code), so a file
            // shorter than that cannot hold the image it announces — rejected before anything is allocated for it.  (+ a cap of 2^28 pixels.)
            {
                const uint64_t blocks = (((uint64_t)width + 7) / 8) * (((uint64_t)height + 7) / 8);
                if ((uint64_t)width * (uint64_t)height > (1ull << 28) || blocks / 8 > data.size()) return false;
            }
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4;
                comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c];
                if (comps[c].tq > 3) return false;
            }
        } else if (marker == 0xC1 || (marker >= 0xC5 && marker <= 0xCF && marker != 0xC8 && marker != 0xCC)) {
            return false;  // extended sequential / lossless / arithmetic coding
        } else if (marker == 0xC4) {
            size_t i = 0;
            while (i + 17 <= seglen) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                if (th > 3 || tc > 1) return false;
                Huff& h = tc ? ac[th] : dc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[i + l]; total += h.bits[l]; }
                i += 17;
                if (total > 256 || i + (size_t)total > seglen) return false;
                std::memcpy(h.vals, seg + i, (size_t)total);
                i += (size_t)total;
                h.build();
                h.present = true;
            }
        } else if (marker == 0xDD) {
            if (seglen < 2) return false;
            restart_interval = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {
            if (comps.empty() || seglen < 1) return false;
            if (progressive) return decodeProgressive(data, pos, qt, dc, ac, comps, restart_interval, width, height, rgb);  // (takes over at this first scan header)
            const int ns = seg[0];
            if (ns != (int)comps.size() || seglen < (size_t)(1 + 2 * ns + 3)) return false;  // one interleaved scan
            for (int k = 0; k < ns; ++k) {
                const int cid = seg[1 + 2 * k];
                Component* cp = nullptr;
                for (auto& c : comps) if (c.id == cid) cp = &c;
                if (!cp) return false;
                cp->td = seg[2 + 2 * k] >> 4;
                cp->ta = seg[2 + 2 * k] & 15;
                if (cp->td > 3 || cp->ta > 3 || !dc[cp->td].present || !ac[cp->ta].present) return false;
            }
            scan = &data[pos + 2 + len];
            break;
        }
        pos += 2 + len;
    }
    if (!scan || comps.empty()) return false;

    int hmax = 1, vmax = 1;
    for (auto& c : comps) {
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2) return false;
        hmax = c.h > hmax ? c.h : hmax;
        vmax = c.v > vmax ? c.v : vmax;
    }
    for (size_t k = 1; k < comps.size(); ++k)
        if (comps[k].h != 1 || comps[k].v != 1) return false;  // chroma at full sampling factor is not supported
    if (comps.size() == 1) { hmax = comps[0].h = 1; vmax = comps[0].v = 1; }
    const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    const int mcux = (width + mcu_w - 1) / mcu_w, mcuy = (height + mcu_h - 1) / mcu_h;
    for (auto& c : comps) {
        c.width = (width * c.h + hmax - 1) / hmax;
        c.height = (height * c.v + vmax - 1) / vmax;
        c.stride = mcux * c.h * 8;
        c.rows = mcuy * c.v * 8;
        c.plane.assign((size_t)c.stride * c.rows, 0);
        c.pred = 0;
    }

    BitReader br{scan, data.data() + data.size()};
    int32_t block[64];
    int restarts_left = restart_interval;
    for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
            if (restart_interval && restarts_left == 0) {
                // skip to the RSTn marker, reset predictors
                const uint8_t* q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                if (q + 1 >= br.end) return false;
                br.p = q + 2;
                br.reset();
                for (auto& c : comps) c.pred = 0;
                restarts_left = restart_interval;
            }
            for (auto& c : comps)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        std::memset(block, 0, sizeof(block));
                        const int t = decodeSymbol(br, dc[c.td]);
                        if (t < 0 || t > 11) return false;
                        const int diff = t ? extend(br.get(t), t) : 0;
                        c.pred += diff;
                        block[0] = c.pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            const int rs = decodeSymbol(br, ac[c.ta]);
                            if (rs < 0) return false;
                            const int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;
                            }
                            k += r;
                            if (k > 63) return false;
                            block[kZigzag[k]] = extend(br.get(s), s) * qt[c.tq][kZigzag[k]];
                            ++k;
                        }
                        uint8_t* dst = c.plane.data() + (size_t)((my * c.v + by) * 8) * c.stride + (size_t)(mx * c.h + bx) * 8;
                        idctIslow(block, dst, c.stride);
                    }
            if (restart_interval) --restarts_left;
        }

    finishImage(comps, hmax, vmax, width, height, rgb);
    return true;
}

}  // namespace trt
