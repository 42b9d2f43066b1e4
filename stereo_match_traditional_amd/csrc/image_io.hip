// Host-side image files for the step before the path: the reference's drivers read
// their pairs with cv::imread(path) and write results with cv::imwrite (AD-CensusV1/main.cpp:16-17,
// :115-117; SADmain.cpp:28-29; ASWeight.cpp:11-12).  No OpenCV, libpng or zlib here: a self-contained
// 8-bit reader for PNG (all five colour types, bit depths 1-16, the five scanline filters, zlib/deflate
// streams with stored, fixed and dynamic blocks; non-interlaced) and binary PGM / PPM, and a writer for
// PNG (filter 0, stored deflate blocks -- valid, uncompressed) and PGM / PPM.  Colour pixels are handed
// out and taken in B, G, R order like cv::Mat.  Pure host code; nothing here touches the GPU.
#include "smt_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

namespace {

// ---- CRC-32 (PNG chunks) and Adler-32 (zlib trailer) ----------------------------------------------
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    struct Table {
        uint32_t v[256];
        Table()
        {
            for (uint32_t k = 0; k < 256; k++) {
                uint32_t c = k;
                for (int b = 0; b < 8; b++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                v[k] = c;
            }
        }
    };
    static const Table tab;                                // C++11: initialised once, thread-safe
    const uint32_t *table = tab.v;
    crc = ~crc;
    for (size_t k = 0; k < n; k++) crc = table[(crc ^ p[k]) & 0xff] ^ (crc >> 8);
    return ~crc;
}
uint32_t adler32(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t k = 0; k < n; k++) { a = (a + p[k]) % 65521u; b = (b + a) % 65521u; }
    return (b << 16) | a;
}

// ---- inflate (RFC 1951) ---------------------------------------------------------------------------
struct BitReader {
    const uint8_t *p; size_t n, pos; uint32_t buf; int cnt; bool bad;
    int bits(int need)
    {
        while (cnt < need) {
            if (pos >= n) { bad = true; return 0; }
            buf |= (uint32_t)p[pos++] << cnt; cnt += 8;
        }
        const int v = (int)(buf & ((1u << need) - 1));
        buf >>= need; cnt -= need;
        return v;
    }
};
struct Huffman { uint16_t count[16]; uint16_t symbol[288]; };

bool build_huffman(Huffman &h, const uint8_t *len, int n)
{
    memset(h.count, 0, sizeof(h.count));
    for (int k = 0; k < n; k++) h.count[len[k]]++;
    if (h.count[0] == n) return true;                      // no codes: legal for an unused distance tree
    int left = 1;
    for (int l = 1; l < 16; l++) { left = (left << 1) - h.count[l]; if (left < 0) return false; }
    uint16_t offs[16]; offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h.count[l];
    for (int k = 0; k < n; k++) if (len[k]) h.symbol[offs[len[k]]++] = (uint16_t)k;
    return true;
}
int decode_symbol(BitReader &br, const Huffman &h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= br.bits(1);
        if (br.bad) return -1;
        const int cnt = h.count[l];
        if (code - cnt < first) return h.symbol[index + (code - first)];
        index += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    return -1;
}
bool inflate_codes(BitReader &br, std::vector<uint8_t> &out, const Huffman &lit, const Huffman &dist, size_t cap)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int sym = decode_symbol(br, lit);
        if (sym < 0) return false;
        if (sym < 256) { if (out.size() >= cap) return false; out.push_back((uint8_t)sym); }
        else if (sym == 256) return true;
        else {
            sym -= 257;
            if (sym >= 29) return false;
            const int len = lbase[sym] + br.bits(lext[sym]);
            const int ds = decode_symbol(br, dist);
            if (ds < 0 || ds >= 30) return false;
            const size_t d = (size_t)dbase[ds] + (size_t)br.bits(dext[ds]);
            if (br.bad || d > out.size() || out.size() + (size_t)len > cap) return false;   // cap: no decompression bombs
            size_t from = out.size() - d;
            for (int k = 0; k < len; k++) out.push_back(out[from + k]);
        }
    }
}
// largest decoded image the reader accepts (65535 x 65535 is what the PNG checks above allow: 17 GB of RGBA16)
constexpr size_t kMaxDecodedBytes = (size_t)1 << 31;

bool zlib_inflate(const uint8_t *p, size_t n, std::vector<uint8_t> &out, size_t cap)
{
    if (n < 6 || (p[0] & 0x0f) != 8 || ((p[0] << 8) | p[1]) % 31 != 0 || (p[1] & 0x20)) return false;
    BitReader br{p + 2, n - 2, 0, 0, 0, false};
    int last;
    do {
        last = br.bits(1);
        const int type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            br.buf = 0; br.cnt = 0;                        // to the byte boundary
            if (br.pos + 4 > br.n) return false;
            const unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n || out.size() + len > cap) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lens[320];
            if (type == 1) {
                int k = 0;
                for (; k < 144; k++) lens[k] = 8;
                for (; k < 256; k++) lens[k] = 9;
                for (; k < 280; k++) lens[k] = 7;
                for (; k < 288; k++) lens[k] = 8;
                build_huffman(lit, lens, 288);
                for (k = 0; k < 30; k++) lens[k] = 5;
                build_huffman(dist, lens, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {};
                for (int k = 0; k < ncode; k++) cl[order[k]] = (uint8_t)br.bits(3);
                Huffman clh;
                if (!build_huffman(clh, cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = decode_symbol(br, clh);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (idx == 0) return false; val = lens[idx - 1]; rep = 3 + br.bits(2); }
                        else if (sym == 17) rep = 3 + br.bits(3);
                        else rep = 11 + br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                if (br.bad || lens[256] == 0) return false;
                if (!build_huffman(lit, lens, nlen) || !build_huffman(dist, lens + nlen, ndist)) return false;
            }
            if (!inflate_codes(br, out, lit, dist, cap)) return false;
        } else return false;
    } while (!last);
    return !br.bad;
}

// ---- PNG --------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// decodes into `channels_out` = 1 (gray stored) or 3 (B, G, R); alpha dropped, 16-bit samples keep their
// high byte, sub-byte gray samples are scaled to 0..255, palettes expanded -- what cv::imread's 8-bit
// paths deliver
int decode_png(const std::vector<uint8_t> &f, std::vector<uint8_t> &pix, int &H, int &W, int &ch)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return SMT_ERR_ARG;
    size_t pos = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool seen_ihdr = false, seen_iend = false;
    while (pos + 12 <= f.size() && !seen_iend) {
        const uint32_t len = be32(&f[pos]);
        if (pos + 12 + (size_t)len > f.size()) return SMT_ERR_ARG;
        const uint8_t *type = &f[pos + 4], *data = &f[pos + 8];
        if (crc32_update(0, type, 4 + (size_t)len) != be32(data + len)) return SMT_ERR_ARG;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return SMT_ERR_ARG;
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return SMT_ERR_ARG;
            seen_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) seen_iend = true;
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || !seen_iend || w == 0 || h == 0 || w > 65535 || h > 65535 || interlace != 0) return SMT_ERR_ARG;
    int spp;                                               // samples per pixel as stored
    switch (ctype) {
    case 0: spp = 1; break;
    case 2: spp = 3; break;
    case 3: spp = 1; break;
    case 4: spp = 2; break;
    case 6: spp = 4; break;
    default: return SMT_ERR_ARG;
    }
    const bool depth_ok = (ctype == 0) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                          : (ctype == 3) ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok || (ctype == 3 && plte.size() < 3)) return SMT_ERR_ARG;
    const size_t bpp = (size_t)(spp * depth + 7) / 8;      // filter distance in bytes
    const size_t stride = ((size_t)w * spp * depth + 7) / 8;
    // a deflate stream expands at most 1032:1, so a file of this size cannot hold more rows than that: refuse headers
    // that promise more before reserving anything, and never inflate past the size the header implies
    const size_t need = (stride + 1) * h;
    if (need > kMaxDecodedBytes || need / 1032 > idat.size() + 1) return SMT_ERR_ARG;
    std::vector<uint8_t> raw;
    raw.reserve(need);
    if (!zlib_inflate(idat.data(), idat.size(), raw, need) || raw.size() < need) return SMT_ERR_ARG;
    // undo the scanline filters in place (row r at raw[r*(stride+1)+1])
    for (uint32_t r = 0; r < h; r++) {
        uint8_t *cur = &raw[(size_t)r * (stride + 1) + 1];
        const uint8_t *up = r ? cur - (stride + 1) : nullptr;
        const int ft = cur[-1];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = cur[x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: return SMT_ERR_ARG;
            }
            cur[x] = (uint8_t)v;
        }
    }
    H = (int)h; W = (int)w;
    ch = (ctype == 0 || ctype == 4) ? 1 : 3;
    pix.assign((size_t)H * W * ch, 0);
    for (int r = 0; r < H; r++) {
        const uint8_t *row = &raw[(size_t)r * (stride + 1) + 1];
        for (int x = 0; x < W; x++) {
            auto sample = [&](int s) -> int {             // s-th sample of pixel x, as 8 bits
                if (depth == 8) return row[(size_t)x * spp + s];
                if (depth == 16) return row[((size_t)x * spp + s) * 2];
                const int idx = x * spp + s, per = 8 / depth;
                const int v = (row[idx / per] >> ((per - 1 - idx % per) * depth)) & ((1 << depth) - 1);
                return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
            };
            uint8_t *o = &pix[((size_t)r * W + x) * ch];
            if (ctype == 0 || ctype == 4) o[0] = (uint8_t)sample(0);
            else if (ctype == 3) {
                const size_t e = (size_t)sample(0) * 3;
                if (e + 3 > plte.size()) return SMT_ERR_ARG;
                o[0] = plte[e + 2]; o[1] = plte[e + 1]; o[2] = plte[e];
            } else { o[0] = (uint8_t)sample(2); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(0); }
        }
    }
    return SMT_OK;
}

int next_int(const std::vector<uint8_t> &f, size_t &pos)
{
    for (;;) {                                             // whitespace and # comments
        while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\t' || f[pos] == '\n' || f[pos] == '\r')) pos++;
        if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; }
        else break;
    }
    int v = -1;
    while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = (v < 0 ? 0 : v) * 10 + (f[pos] - '0'); pos++; if (v > (1 << 24)) return -1; }
    return v;
}
int decode_pnm(const std::vector<uint8_t> &f, std::vector<uint8_t> &pix, int &H, int &W, int &ch)
{
    if (f.size() < 7 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return SMT_ERR_ARG;
    ch = f[1] == '5' ? 1 : 3;
    size_t pos = 2;
    W = next_int(f, pos); H = next_int(f, pos);
    const int maxv = next_int(f, pos);
    if (W <= 0 || H <= 0 || maxv <= 0 || maxv > 65535 || pos >= f.size()) return SMT_ERR_ARG;
    pos++;                                                 // the single whitespace byte after maxval
    const size_t bps = maxv > 255 ? 2 : 1, need = (size_t)H * W * ch * bps;
    if (pos + need > f.size()) return SMT_ERR_ARG;
    pix.resize((size_t)H * W * ch);
    for (size_t p = 0; p < (size_t)H * W; p++)
        for (int c = 0; c < ch; c++) {
            const size_t s = (p * ch + c) * bps;
            const int v = bps == 2 ? ((f[pos + s] << 8) | f[pos + s + 1]) : f[pos + s];
            pix[p * ch + (ch == 3 ? 2 - c : c)] = (uint8_t)(maxv == 255 ? v : (bps == 2 ? v >> 8 : v * 255 / maxv));   // RGB file -> BGR
        }
    return SMT_OK;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void put_chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data)
{
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32_update(0, &out[at], 4 + data.size()));
}

bool has_ext(const char *path, const char *ext)
{
    const size_t n = strlen(path), m = strlen(ext);
    if (n < m) return false;
    for (size_t k = 0; k < m; k++) {
        char c = path[n - m + k];
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
        if (c != ext[k]) return false;
    }
    return true;
}

}  // namespace

static int image_read_impl(const char *path, int want_channels, uint8_t **pixels, int *H, int *W, int *channels)
{
    if (!path || !pixels || !H || !W || !channels || (want_channels != 0 && want_channels != 1 && want_channels != 3))
        return SMT_ERR_ARG;
    FILE *fp = fopen(path, "rb");
    if (!fp) return SMT_ERR_ARG;
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    std::vector<uint8_t> pix;
    int h = 0, w = 0, ch = 0;
    int rc = (f.size() >= 2 && f[0] == 'P') ? decode_pnm(f, pix, h, w, ch) : decode_png(f, pix, h, w, ch);
    if (rc != SMT_OK) return rc;
    const int outc = want_channels ? want_channels : ch;
    uint8_t *o = (uint8_t *)malloc((size_t)h * w * outc);
    if (!o) return SMT_ERR_ALLOC;
    for (size_t p = 0; p < (size_t)h * w; p++) {
        if (outc == ch) memcpy(o + p * outc, &pix[p * ch], (size_t)ch);
        else if (outc == 3) o[p * 3] = o[p * 3 + 1] = o[p * 3 + 2] = pix[p];              // gray file as colour: replicated
        else {                                                                            // colour file as gray: cvtColor's BGR2GRAY rule
            const uint8_t *s = &pix[p * 3];
            o[p] = (uint8_t)((1868 * s[0] + 9617 * s[1] + 4899 * s[2] + 8192) >> 14);
        }
    }
    *pixels = o; *H = h; *W = w; *channels = outc;
    return SMT_OK;
}

// The codec parses untrusted files behind a C ABI: no C++ exception may cross it.
SMT_API int smt_image_read(const char *path, int want_channels, uint8_t **pixels, int *H, int *W, int *channels)
{
    try {
        return image_read_impl(path, want_channels, pixels, H, W, channels);
    } catch (const std::bad_alloc &) {
        return SMT_ERR_ALLOC;
    } catch (...) {
        return SMT_ERR_ARG;
    }
}

SMT_API int smt_image_free(uint8_t *pixels)
{
    free(pixels);
    return SMT_OK;
}

static int image_write_impl(const char *path, const uint8_t *pixels, int H, int W, int channels)
{
    if (!path || !pixels || H <= 0 || W <= 0 || H > 65535 || W > 65535 || (channels != 1 && channels != 3)) return SMT_ERR_ARG;
    std::vector<uint8_t> out;
    const size_t stride = (size_t)W * channels;
    if (has_ext(path, ".pgm") || has_ext(path, ".ppm") || has_ext(path, ".pnm")) {
        char hdr[64];
        const int hl = snprintf(hdr, sizeof(hdr), "P%d\n%d %d\n255\n", channels == 1 ? 5 : 6, W, H);
        out.insert(out.end(), hdr, hdr + hl);
        for (size_t p = 0; p < (size_t)H * W; p++)
            for (int c = 0; c < channels; c++) out.push_back(pixels[p * channels + (channels == 3 ? 2 - c : c)]);
    } else if (has_ext(path, ".png")) {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        out.insert(out.end(), sig, sig + 8);
        std::vector<uint8_t> ihdr;
        put_be32(ihdr, (uint32_t)W); put_be32(ihdr, (uint32_t)H);
        ihdr.push_back(8); ihdr.push_back(channels == 1 ? 0 : 2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
        put_chunk(out, "IHDR", ihdr);
        std::vector<uint8_t> raw;                          // filter byte 0 + row, RGB order in the file
        raw.reserve((stride + 1) * H);
        for (int r = 0; r < H; r++) {
            raw.push_back(0);
            for (int x = 0; x < W; x++)
                for (int c = 0; c < channels; c++) raw.push_back(pixels[((size_t)r * W + x) * channels + (channels == 3 ? 2 - c : c)]);
        }
        std::vector<uint8_t> z;                            // zlib stream of stored blocks
        z.push_back(0x78); z.push_back(0x01);
        size_t pos = 0;
        do {
            const size_t len = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
            z.push_back(pos + len == raw.size() ? 1 : 0);
            z.push_back(len & 0xff); z.push_back(len >> 8); z.push_back(~len & 0xff); z.push_back((~len >> 8) & 0xff);
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
            pos += len;
        } while (pos < raw.size());
        put_be32(z, adler32(raw.data(), raw.size()));
        put_chunk(out, "IDAT", z);
        put_chunk(out, "IEND", std::vector<uint8_t>());
    } else return SMT_ERR_ARG;
    FILE *fp = fopen(path, "wb");
    if (!fp) return SMT_ERR_ARG;
    const bool ok = fwrite(out.data(), 1, out.size(), fp) == out.size();
    return (fclose(fp) == 0 && ok) ? SMT_OK : SMT_ERR_ARG;
}

SMT_API int smt_image_write(const char *path, const uint8_t *pixels, int H, int W, int channels)
{
    try {
        return image_write_impl(path, pixels, H, W, channels);
    } catch (const std::bad_alloc &) {
        return SMT_ERR_ALLOC;
    } catch (...) {
        return SMT_ERR_ARG;
    }
}
