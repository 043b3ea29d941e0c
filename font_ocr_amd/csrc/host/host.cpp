// host.cpp — CPU-side host utilities of the ncc path (libfocr_host.so):
// bank files, image decode (image::open(..).into_luma8(), src/ncc.rs:575),
// synthetic pages and Rust-compatible float printing
// for the --csv / --raw formats (src/ncc.rs:685-697, 855-864).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "focr_host.h"

namespace {

int fail(char *err, size_t errlen, const std::string &msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
    return 1;
}

// image 0.25 rgb -> luma (sRGB weights 2126/7152/722 over 10000)
inline uint8_t rgb_to_luma(uint32_t r, uint32_t g, uint32_t b) {
    return (uint8_t)((2126 * r + 7152 * g + 722 * b) / 10000);
}
inline uint8_t u16_to_u8(uint32_t v) { return (uint8_t)((v + 128) / 257); }

bool read_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) {
        fclose(f);
        return false;
    }
    buf.resize((size_t)n);
    size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

// ---- PNM ---------------------------------------------------------------
struct PnmTok {
    const uint8_t *p, *e;
    bool next_uint(uint32_t *v) {
        for (;;) {
            while (p < e && isspace(*p)) p++;
            if (p < e && *p == '#') {
                while (p < e && *p != '\n') p++;
                continue;
            }
            break;
        }
        if (p >= e || !isdigit(*p)) return false;
        uint64_t acc = 0;
        while (p < e && isdigit(*p)) {
            acc = acc * 10 + (*p - '0');
            if (acc > 0xffffffffu) return false;
            p++;
        }
        *v = (uint32_t)acc;
        return true;
    }
};

int load_pnm(const std::vector<uint8_t> &d, uint8_t **px, size_t *w, size_t *h, char *err, size_t errlen) {
    int kind = d[1] - '0';
    PnmTok t{d.data() + 2, d.data() + d.size()};
    uint32_t W, H, maxv = 1;
    if (!t.next_uint(&W) || !t.next_uint(&H)) return fail(err, errlen, "pnm: bad header");
    if (kind != 1 && kind != 4 && !t.next_uint(&maxv)) return fail(err, errlen, "pnm: bad maxval");
    if (W == 0 || H == 0 || maxv == 0 || maxv > 65535) return fail(err, errlen, "pnm: bad dimensions");
    size_t npx = (size_t)W * H;
    uint8_t *out = (uint8_t *)malloc(npx);
    if (!out) return fail(err, errlen, "out of memory");
    const int ch = (kind == 3 || kind == 6) ? 3 : 1;
    auto conv = [&](uint32_t v) -> uint32_t { return maxv > 255 ? u16_to_u8((uint32_t)((uint64_t)v * 65535 / maxv)) : v; };
    if (kind == 1 || kind == 2 || kind == 3) {  // ASCII
        for (size_t i = 0; i < npx; i++) {
            uint32_t c[3] = {0, 0, 0};
            for (int k = 0; k < ch; k++) {
                if (kind == 1) {  // bits may be unseparated
                    while (t.p < t.e && (isspace(*t.p))) t.p++;
                    if (t.p >= t.e) { free(out); return fail(err, errlen, "pnm: truncated"); }
                    c[k] = (*t.p++ == '1') ? 0 : 255;
                } else {
                    if (!t.next_uint(&c[k])) { free(out); return fail(err, errlen, "pnm: truncated"); }
                    c[k] = conv(c[k]);
                }
            }
            out[i] = ch == 3 ? rgb_to_luma(c[0], c[1], c[2]) : (uint8_t)c[0];
        }
    } else {
        if (t.p >= t.e) { free(out); return fail(err, errlen, "pnm: truncated"); }
        t.p++;  // single whitespace after the header
        const uint8_t *s = t.p;
        size_t avail = (size_t)(t.e - t.p);
        if (kind == 4) {
            size_t stride = (W + 7) / 8;
            if (avail < stride * H) { free(out); return fail(err, errlen, "pnm: truncated"); }
            for (size_t y = 0; y < H; y++)
                for (size_t x = 0; x < W; x++)
                    out[y * W + x] = ((s[y * stride + x / 8] >> (7 - x % 8)) & 1) ? 0 : 255;
        } else {
            size_t bps = maxv > 255 ? 2 : 1;
            if (avail < npx * ch * bps) { free(out); return fail(err, errlen, "pnm: truncated"); }
            for (size_t i = 0; i < npx; i++) {
                uint32_t c[3] = {0, 0, 0};
                for (int k = 0; k < ch; k++) {
                    const uint8_t *q = s + (i * ch + k) * bps;
                    c[k] = conv(bps == 2 ? ((uint32_t)q[0] << 8 | q[1]) : q[0]);
                }
                out[i] = ch == 3 ? rgb_to_luma(c[0], c[1], c[2]) : (uint8_t)c[0];
            }
        }
    }
    *px = out;
    *w = W;
    *h = H;
    return 0;
}

// ---- PNG (non-interlaced and Adam7; gray/rgb/palette, 1-16 bit) ----------
inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | p[1] << 16 | p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// un-filter `rows` scanlines of `rowbytes` (+1 filter byte each) in place; bpp = bytes per complete pixel (>=1)
bool unfilter(uint8_t *data, size_t rows, size_t rowbytes, size_t bpp) {
    std::vector<uint8_t> zero(rowbytes, 0);
    const uint8_t *prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        uint8_t *line = data + y * (rowbytes + 1);
        uint8_t ft = line[0];
        uint8_t *cur = line + 1;
        for (size_t i = 0; i < rowbytes; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int add;
            switch (ft) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = b; break;
                case 3: add = (a + b) >> 1; break;
                case 4: add = paeth(a, b, c); break;
                default: return false;
            }
            cur[i] = (uint8_t)(cur[i] + add);
        }
        prev = cur;
    }
    return true;
}

int load_png(const std::vector<uint8_t> &d, uint8_t **px, size_t *w, size_t *h, char *err, size_t errlen) {
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]);
        const uint8_t *type = &d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return fail(err, errlen, "png: truncated chunk");
        const uint8_t *body = &d[pos + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            W = be32(body);
            H = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || W == 0 || H == 0) return fail(err, errlen, "png: missing IHDR");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return fail(err, errlen, "png: bad colour type");
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return fail(err, errlen, "png: bad bit depth");
    const size_t bits_pp = (size_t)depth * channels;
    const size_t bpp = std::max<size_t>(1, bits_pp / 8);

    struct Pass { size_t x0, y0, dx, dy; };
    std::vector<Pass> passes;
    if (interlace == 0) passes.push_back({0, 0, 1, 1});
    else if (interlace == 1)
        passes = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    else return fail(err, errlen, "png: bad interlace method");

    size_t total = 0;
    for (auto &p : passes) {
        size_t pw = W > p.x0 ? (W - p.x0 + p.dx - 1) / p.dx : 0, ph = H > p.y0 ? (H - p.y0 + p.dy - 1) / p.dy : 0;
        if (pw && ph) total += ph * ((pw * bits_pp + 7) / 8 + 1);
    }
    std::vector<uint8_t> raw(total);
    z_stream zs{};
    if (inflateInit(&zs) != Z_OK) return fail(err, errlen, "png: zlib init failed");
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    int zr = inflate(&zs, Z_FINISH);
    size_t produced = raw.size() - zs.avail_out;
    inflateEnd(&zs);
    if ((zr != Z_STREAM_END && zr != Z_OK && zr != Z_BUF_ERROR) || produced < total) return fail(err, errlen, "png: inflate failed");

    size_t npx = (size_t)W * H;
    uint8_t *out = (uint8_t *)malloc(npx);
    if (!out) return fail(err, errlen, "out of memory");
    size_t off = 0;
    for (auto &p : passes) {
        size_t pw = W > p.x0 ? (W - p.x0 + p.dx - 1) / p.dx : 0, ph = H > p.y0 ? (H - p.y0 + p.dy - 1) / p.dy : 0;
        if (!pw || !ph) continue;
        size_t rowbytes = (pw * bits_pp + 7) / 8;
        if (!unfilter(raw.data() + off, ph, rowbytes, bpp)) { free(out); return fail(err, errlen, "png: bad filter"); }
        for (size_t y = 0; y < ph; y++) {
            const uint8_t *line = raw.data() + off + y * (rowbytes + 1) + 1;
            for (size_t x = 0; x < pw; x++) {
                uint32_t c[4] = {0, 0, 0, 0};
                for (int k = 0; k < channels; k++) {
                    if (depth == 8) c[k] = line[x * channels + k];
                    else if (depth == 16) c[k] = u16_to_u8((uint32_t)line[(x * channels + k) * 2] << 8 | line[(x * channels + k) * 2 + 1]);
                    else {
                        size_t bit = x * depth;
                        uint32_t v = (line[bit / 8] >> (8 - depth - bit % 8)) & ((1u << depth) - 1);
                        c[k] = ctype == 3 ? v : v * 255 / ((1u << depth) - 1);
                    }
                }
                uint8_t l;
                if (ctype == 3) {
                    size_t idx = c[0];
                    if (idx * 3 + 2 < plte.size()) l = rgb_to_luma(plte[idx * 3], plte[idx * 3 + 1], plte[idx * 3 + 2]);
                    else l = 0;
                } else if (ctype == 2 || ctype == 6) l = rgb_to_luma(c[0], c[1], c[2]);
                else l = (uint8_t)c[0];
                out[(p.y0 + y * p.dy) * W + (p.x0 + x * p.dx)] = l;
            }
        }
        off += ph * (rowbytes + 1);
    }
    *px = out;
    *w = W;
    *h = H;
    return 0;
}

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

const char BANK_MAGIC[8] = {'F', 'O', 'C', 'R', 'B', 'N', 'K', '1'};

}  // namespace

extern "C" {

void focr_bank_free(focr_bank_t *bank) {
    if (!bank) return;
    free(bank->templates);
    free(bank->needles);
    memset(bank, 0, sizeof(*bank));
}

int focr_bank_save(const char *path, const focr_bank_t *b) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    uint32_t hdr[4] = {(uint32_t)b->n_templates, b->n_alphabet, b->x_bits, b->y_bits};
    float fl[2] = {b->text_size, b->advance_px};
    uint64_t nl = b->needles_len;
    bool ok = fwrite(BANK_MAGIC, 1, 8, f) == 8 && fwrite(hdr, 4, 4, f) == 4 && fwrite(fl, 4, 2, f) == 2 &&
              fwrite(&nl, 8, 1, f) == 1 &&
              fwrite(b->templates, sizeof(focr_template_t), b->n_templates, f) == b->n_templates &&
              fwrite(b->needles, 1, b->needles_len, f) == b->needles_len;
    fclose(f);
    return ok ? 0 : 1;
}

int focr_bank_load(const char *path, focr_bank_t *out) {
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    char magic[8];
    uint32_t hdr[4];
    float fl[2];
    uint64_t nl;
    bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, BANK_MAGIC, 8) && fread(hdr, 4, 4, f) == 4 &&
              fread(fl, 4, 2, f) == 2 && fread(&nl, 8, 1, f) == 1 && hdr[0] <= (1u << 24) && nl <= (1ull << 32);
    if (ok) {
        out->n_templates = hdr[0];
        out->n_alphabet = hdr[1];
        out->x_bits = hdr[2];
        out->y_bits = hdr[3];
        out->text_size = fl[0];
        out->advance_px = fl[1];
        out->needles_len = (size_t)nl;
        out->templates = (focr_template_t *)malloc(sizeof(focr_template_t) * (hdr[0] ? hdr[0] : 1));
        out->needles = (uint8_t *)malloc(nl ? (size_t)nl : 1);
        ok = out->templates && out->needles &&
             fread(out->templates, sizeof(focr_template_t), hdr[0], f) == hdr[0] &&
             fread(out->needles, 1, (size_t)nl, f) == (size_t)nl;
        for (size_t i = 0; ok && i < out->n_templates; i++) {
            const focr_template_t &t = out->templates[i];
            ok = (uint64_t)t.offset + (uint64_t)t.n_w * t.n_h <= nl;
        }
    }
    fclose(f);
    if (!ok) focr_bank_free(out);
    return ok ? 0 : 1;
}

int focr_image_load_luma8(const char *path, uint8_t **px, size_t *w, size_t *h, char *err, size_t errlen) {
    std::vector<uint8_t> d;
    if (!read_file(path, d)) return fail(err, errlen, std::string("cannot read ") + path);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (d.size() >= 8 && !memcmp(d.data(), png_sig, 8)) return load_png(d, px, w, h, err, errlen);
    if (d.size() >= 3 && d[0] == 'P' && d[1] >= '1' && d[1] <= '6') return load_pnm(d, px, w, h, err, errlen);
    return fail(err, errlen, "unsupported image format (pnm and png only, Cargo.toml:10)");
}

// Size of an image from the first bytes of its file (PNM header / PNG IHDR); 0 on success.
int focr_image_probe(const char *path, size_t *w, size_t *h, char *err, size_t errlen) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(err, errlen, std::string("cannot read ") + path);
    uint8_t hd[512];
    const size_t n = fread(hd, 1, sizeof hd, f);
    fclose(f);
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n >= 24 && !memcmp(hd, png_sig, 8)) {  // IHDR is the first chunk: width, height big-endian at 16, 20
        *w = ((size_t)hd[16] << 24) | ((size_t)hd[17] << 16) | ((size_t)hd[18] << 8) | hd[19];
        *h = ((size_t)hd[20] << 24) | ((size_t)hd[21] << 16) | ((size_t)hd[22] << 8) | hd[23];
        return (*w && *h) ? 0 : fail(err, errlen, "png: bad header");
    }
    if (n >= 3 && hd[0] == 'P' && hd[1] >= '1' && hd[1] <= '6') {
        PnmTok t{hd + 2, hd + n};
        uint32_t W, H;
        if (!t.next_uint(&W) || !t.next_uint(&H) || !W || !H) return fail(err, errlen, "pnm: bad header");
        *w = W;
        *h = H;
        return 0;
    }
    return fail(err, errlen, "unsupported image format (pnm and png only, Cargo.toml:10)");
}

// Decode straight into caller memory (e.g. a page-locked batch slab): binary 8-bit PGM is one fread into dst, every
// other format goes through the general decoder and one copy.  1 if the image does not fit `cap` bytes or cannot be read.
int focr_image_load_luma8_into(const char *path, uint8_t *dst, size_t cap, size_t *w, size_t *h, char *err, size_t errlen) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail(err, errlen, std::string("cannot read ") + path);
    uint8_t hd[512];
    const size_t n = fread(hd, 1, sizeof hd, f);
    if (n >= 3 && hd[0] == 'P' && hd[1] == '5') {
        PnmTok t{hd + 2, hd + n};
        uint32_t W, H, maxv;
        if (t.next_uint(&W) && t.next_uint(&H) && t.next_uint(&maxv) && W && H && maxv == 255 && t.p < t.e) {
            const size_t hdr = (size_t)(t.p + 1 - hd), npx = (size_t)W * H;  // a single whitespace byte ends the header
            if (npx > cap) {
                fclose(f);
                return fail(err, errlen, "image larger than its slot");
            }
            const size_t have = std::min(npx, n - std::min(n, hdr));
            memcpy(dst, hd + hdr, have);
            const size_t got = have + (npx > have ? fread(dst + have, 1, npx - have, f) : 0);
            fclose(f);
            if (got != npx) return fail(err, errlen, "pnm: truncated");
            *w = W;
            *h = H;
            return 0;
        }
    }
    fclose(f);
    uint8_t *px = nullptr;
    if (int rc = focr_image_load_luma8(path, &px, w, h, err, errlen)) return rc;
    int rc = 0;
    if (*w * *h > cap) rc = fail(err, errlen, "image larger than its slot");
    else memcpy(dst, px, *w * *h);
    free(px);
    return rc;
}

size_t focr_line_text(const focr_hit_t *chars, size_t n, float advance_px, int spaces, char *out, size_t cap) {
    size_t need = 0;
    auto put = [&](char ch) {
        if (out && need + 1 < cap) out[need] = ch;
        need++;
    };
    for (size_t i = 0; i < n; i++) {
        if (spaces && i > 0 && advance_px > 0.f) {
            const float dx = (float)chars[i].x - (float)chars[i - 1].x;
            const long cells = std::lround(dx / advance_px);
            for (long k = 1; k < cells; k++) put(' ');
        }
        const uint32_t cp = chars[i].letter;
        if (cp < 0x80) {
            put((char)cp);
        } else if (cp < 0x800) {
            put((char)(0xc0 | (cp >> 6)));
            put((char)(0x80 | (cp & 0x3f)));
        } else if (cp < 0x10000) {
            put((char)(0xe0 | (cp >> 12)));
            put((char)(0x80 | ((cp >> 6) & 0x3f)));
            put((char)(0x80 | (cp & 0x3f)));
        } else {
            put((char)(0xf0 | (cp >> 18)));
            put((char)(0x80 | ((cp >> 12) & 0x3f)));
            put((char)(0x80 | ((cp >> 6) & 0x3f)));
            put((char)(0x80 | (cp & 0x3f)));
        }
    }
    if (out && cap) out[std::min(need, cap - 1)] = 0;
    return need;
}

int focr_image_save_pgm(const char *path, const uint8_t *px, size_t w, size_t h) {
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fprintf(f, "P5\n%zu %zu\n255\n", w, h);
    bool ok = fwrite(px, 1, w * h, f) == w * h;
    fclose(f);
    return ok ? 0 : 1;
}

size_t focr_synth_page(const focr_bank_t *bank, uint64_t seed, size_t r_w, size_t r_h, uint8_t *luma_out,
                       focr_hit_t *truth, size_t truth_cap) {
    std::vector<uint8_t> ink(r_w * r_h, 0);
    size_t stamped = 0;
    const size_t A = bank->n_alphabet, nx = (size_t)1 << bank->x_bits, ny = (size_t)1 << bank->y_bits;
    std::vector<uint32_t> live;  // alphabet indices whose shift-0 raster has ink
    for (size_t a = 0; a < A && a < bank->n_templates; a++) {
        const focr_template_t &t = bank->templates[a];
        uint32_t s = 0;
        for (size_t i = 0; i < (size_t)t.n_w * t.n_h; i++) s += bank->needles[t.offset + i];
        if (s) live.push_back((uint32_t)a);
    }
    if (!live.empty() && bank->n_templates >= A * nx * ny && bank->advance_px > 0.f) {
        SplitMix64 rng{seed};
        const size_t margin = 45, top = 39;
        const size_t line_h = bank->templates[0].n_h ? bank->templates[0].n_h : 1;
        size_t line = 0;
        for (size_t y = top; y + line_h + top <= r_h; y += line_h, line++) {
            const size_t sy = line % ny;
            double pen = (double)margin;
            for (;;) {
                size_t ix = (size_t)std::floor(pen);
                double frac = pen - (double)ix;
                size_t sx = std::min(nx - 1, (size_t)std::floor(frac * (double)nx));
                uint32_t a = live[rng.below((uint32_t)live.size())];
                size_t ti = (sx * ny + sy) * A + a;
                const focr_template_t &t = bank->templates[ti];
                if (ix + t.n_w + margin > r_w) break;
                if (y + t.n_h > r_h) break;
                const uint8_t *src = bank->needles + t.offset;
                for (size_t j = 0; j < t.n_h; j++)
                    for (size_t i = 0; i < t.n_w; i++) {
                        uint8_t &dst = ink[(y + j) * r_w + ix + i];
                        dst = (uint8_t)std::min<uint32_t>(255, (uint32_t)dst + src[j * t.n_w + i]);
                    }
                if (truth && stamped < truth_cap) {
                    focr_hit_t &h = truth[stamped];
                    h.x = (uint16_t)ix;
                    h.y = (uint16_t)y;
                    h.w = t.n_w;
                    h.h = t.n_h;
                    h.similarity = 1.f;
                    h.letter = t.letter;
                    h.template_index = (uint32_t)ti;
                }
                stamped++;
                pen += (double)bank->advance_px;
            }
        }
    }
    for (size_t i = 0; i < r_w * r_h; i++) luma_out[i] = (uint8_t)(255 - ink[i]);
    return stamped;
}

size_t focr_format_f32(float v, char *buf, size_t buflen) {
    if (!buflen) return 0;
    char tmp[128];
    size_t n;
    if (std::isnan(v)) n = (size_t)snprintf(tmp, sizeof tmp, "NaN");
    else if (std::isinf(v)) n = (size_t)snprintf(tmp, sizeof tmp, v < 0 ? "-inf" : "inf");
    else {
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed);
        n = (size_t)(r.ptr - tmp);
    }
    n = std::min(n, buflen - 1);
    memcpy(buf, tmp, n);
    buf[n] = 0;
    return n;
}

}  // extern "C"
