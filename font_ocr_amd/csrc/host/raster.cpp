// raster.cpp — CPU rasterisation of the glyph-template bank (libfocr_raster.so).
//
// Follows get_hits' bank loop and render() of the reference
// (src/ncc.rs:563-573 offsets, 587-628 box size, 629-641 per-letter render,
// 143-196 render) on top of FreeType directly.  The reference goes through
// font-kit 0.14 (freetype loader) + pathfinder_geometry 0.5, neither of which
// is vendored in /root/reference; their behaviour is restated from their
// published semantics (DESIGN.md "third-party arithmetic"):
//   * typographic_bounds: FT_Load_Glyph(NO_HINTING) at char size = units_per_em,
//     rect (horiBearingX, horiBearingY - height, width, height) / 64.
//   * raster_bounds: typographic bounds * (size / upem), y flipped to
//     top-left origin, translated, round_out.
//   * rasterize_glyph: FT_Set_Transform(identity, delta = trunc(t * 64) with y
//     negated), FT_Set_Char_Size(size * 64), FT_Load_Glyph(RENDER | flags),
//     copy the 8-bit bitmap at (bitmap_left, -bitmap_top), clipped to the canvas.
// Parity of the rasterised bytes with the reference is unpinned (no fixtures
// upstream); the scan's parity contract is on identical (page, bank) bytes.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <ft2build.h>
#include FT_FREETYPE_H
#include FT_TRUETYPE_TABLES_H

#include "focr_host.h"

namespace {

struct RectF {
    float ox = 0, oy = 0, lx = 0, ly = 0;  // origin, lower-right (pathfinder RectF layout)
    float width() const { return lx - ox; }
    float height() const { return ly - oy; }
};

RectF scale(const RectF &r, float f) { return {r.ox * f, r.oy * f, r.lx * f, r.ly * f}; }
RectF union_rect(const RectF &a, const RectF &b) {
    return {std::fmin(a.ox, b.ox), std::fmin(a.oy, b.oy), std::fmax(a.lx, b.lx), std::fmax(a.ly, b.ly)};
}
struct RectI {
    int ox, oy, lx, ly;
    int width() const { return lx - ox; }
    int height() const { return ly - oy; }
};
RectI round_out(const RectF &r) {
    return {(int)std::floor(r.ox), (int)std::floor(r.oy), (int)std::ceil(r.lx), (int)std::ceil(r.ly)};
}

struct Face {
    FT_Library lib = nullptr;
    FT_Face face = nullptr;
    ~Face() {
        if (face) FT_Done_Face(face);
        if (lib) FT_Done_FreeType(lib);
    }
    void reset_size() {  // font-kit keeps the face at char size = units_per_em between calls
        FT_Set_Char_Size(face, (FT_F26Dot6)face->units_per_EM << 6, 0, 0, 0);
        FT_Set_Transform(face, nullptr, nullptr);
    }
    bool typographic_bounds(FT_UInt gid, RectF *out) {
        reset_size();
        if (FT_Load_Glyph(face, gid, FT_LOAD_DEFAULT | FT_LOAD_NO_HINTING) != 0) return false;
        const FT_Glyph_Metrics &m = face->glyph->metrics;
        int ox = (int)m.horiBearingX, oy = (int)(m.horiBearingY - m.height);
        int w = (int)m.width, h = (int)m.height;
        out->ox = ox / 64.0f;
        out->oy = oy / 64.0f;
        out->lx = (ox + w) / 64.0f;
        out->ly = (oy + h) / 64.0f;
        return true;
    }
    // Loader::raster_bounds default implementation for a pure translation.
    bool raster_bounds(FT_UInt gid, float size, float tx, float ty, RectI *out) {
        RectF tb;
        if (!typographic_bounds(gid, &tb)) return false;
        RectF trb = scale(tb, size / (float)face->units_per_EM);
        float nox = trb.ox, noy = -trb.oy - trb.height();
        RectF r{nox + tx, noy + ty, nox + trb.width() + tx, noy + trb.height() + ty};
        *out = round_out(r);
        return true;
    }
};

int fail(char *err, size_t errlen, const char *msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg);
    return 1;
}

}  // namespace

extern "C" int focr_raster_bank(const char *font_path, float text_size, uint32_t x_bits,
                                uint32_t y_bits, int hinting, const uint32_t *alphabet,
                                size_t n_alphabet, int box_size, uint32_t x_padding,
                                uint32_t y_padding, focr_bank_t *out, char *err, size_t errlen) {
    if (!font_path || !alphabet || !n_alphabet || !out) return fail(err, errlen, "bad arguments");
    if (x_bits > 8 || y_bits > 8) return fail(err, errlen, "x_bits / y_bits too large");
    Face f;
    if (FT_Init_FreeType(&f.lib) != 0) return fail(err, errlen, "FT_Init_FreeType failed");
    if (FT_New_Face(f.lib, font_path, 0, &f.face) != 0) return fail(err, errlen, "cannot open font");
    const float upem = (float)f.face->units_per_EM;
    const float to_px = (1.f / upem) * text_size;  // src/ncc.rs:579

    std::vector<FT_UInt> gids(n_alphabet);
    for (size_t i = 0; i < n_alphabet; i++) {
        gids[i] = FT_Get_Char_Index(f.face, alphabet[i]);
        if (gids[i] == 0) return fail(err, errlen, "alphabet character missing from font");  // unwrap() at :154
    }

    const size_t nx = (size_t)1 << x_bits, ny = (size_t)1 << y_bits;
    const float x_div = 1.f / (float)nx, y_div = 1.f / (float)ny;  // src/ncc.rs:565-566

    std::vector<focr_template_t> templates;
    std::vector<uint8_t> needles;

    for (size_t sx = 0; sx < nx; sx++)
        for (size_t sy = 0; sy < ny; sy++) {  // x-major, src/ncc.rs:567-571
            const float off[2] = {(float)sx * x_div, (float)sy * y_div};
            float y_offset = 0.f;
            bool have_size = false;
            int cw = 0, ch = 0;
            if (box_size == FOCR_BOX_FONT) {  // src/ncc.rs:589-599
                RectF bbox{(float)f.face->bbox.xMin, (float)f.face->bbox.yMin, (float)f.face->bbox.xMax,
                           (float)f.face->bbox.yMax};
                RectI r = round_out(scale(bbox, to_px));
                cw = r.width();
                ch = r.height();
                y_offset = std::ceil((float)f.face->ascender * to_px);
                have_size = true;
            } else if (box_size == FOCR_BOX_ALPHABET) {  // src/ncc.rs:600-626
                RectF bounds;
                for (size_t i = 0; i < n_alphabet; i++) {
                    RectF tb;
                    if (!f.typographic_bounds(gids[i], &tb)) return fail(err, errlen, "glyph load failed");
                    RectF gb = scale(tb, to_px);
                    float bearing_y = gb.oy + gb.height();
                    RectI rr;
                    if (!f.raster_bounds(gids[i], text_size, off[0], off[1], &rr))
                        return fail(err, errlen, "glyph load failed");
                    y_offset = std::fmax(y_offset, std::ceil(bearing_y));
                    bounds = union_rect(bounds, RectF{(float)rr.ox, (float)rr.oy, (float)rr.lx, (float)rr.ly});
                }
                RectI r = round_out(bounds);
                cw = r.width();
                ch = r.height();
                have_size = true;
            }
            const float corrected[2] = {off[0], off[1] + y_offset};  // src/ncc.rs:629

            for (size_t i = 0; i < n_alphabet; i++) {  // render(), src/ncc.rs:143-196
                RectI rb;
                if (!f.raster_bounds(gids[i], text_size, corrected[0], corrected[1], &rb))
                    return fail(err, errlen, "glyph load failed");
                int w = (have_size ? cw : rb.width()) + 2 * (int)x_padding;
                int h = (have_size ? ch : rb.height()) + 2 * (int)y_padding;
                float origin_x = have_size ? 0.f : -(float)rb.ox;
                float origin_y = have_size ? 0.f : -(float)rb.oy;
                if (w < 0) w = 0;
                if (h < 0) h = 0;
                if (w > 0xffff || h > 0xffff) return fail(err, errlen, "canvas too large");
                std::vector<uint8_t> canvas((size_t)w * (size_t)h, 0);

                // transform vector = origin + padding + pos
                float tvx = origin_x + (float)x_padding + corrected[0];
                float tvy = origin_y + (float)y_padding + corrected[1];
                FT_Vector delta;
                delta.x = (FT_Pos)(int32_t)(tvx * 64.0f);
                delta.y = -(FT_Pos)(int32_t)(tvy * 64.0f);
                FT_Matrix shape{65536, 0, 0, 65536};
                FT_Set_Transform(f.face, &shape, &delta);
                if (FT_Set_Char_Size(f.face, (FT_F26Dot6)(int32_t)(text_size * 64.0f), 0, 0, 0) != 0)
                    return fail(err, errlen, "FT_Set_Char_Size failed");
                FT_Int32 flags = FT_LOAD_DEFAULT | FT_LOAD_RENDER;
                flags |= hinting ? FT_LOAD_TARGET_NORMAL : (FT_LOAD_TARGET_NORMAL | FT_LOAD_NO_HINTING);
                if (FT_Load_Glyph(f.face, gids[i], flags) != 0) return fail(err, errlen, "FT_Load_Glyph failed");
                const FT_GlyphSlot slot = f.face->glyph;
                const FT_Bitmap &bm = slot->bitmap;
                if (bm.buffer && bm.width && bm.rows) {
                    if (bm.pixel_mode != FT_PIXEL_MODE_GRAY) return fail(err, errlen, "unexpected pixel mode");
                    int dx = slot->bitmap_left, dy = -slot->bitmap_top;
                    for (int y = 0; y < (int)bm.rows; y++) {
                        int cy = dy + y;
                        if (cy < 0 || cy >= h) continue;
                        const uint8_t *src = bm.buffer + (ptrdiff_t)y * bm.pitch;
                        for (int x = 0; x < (int)bm.width; x++) {
                            int cx = dx + x;
                            if (cx < 0 || cx >= w) continue;
                            canvas[(size_t)cy * w + cx] = src[x];
                        }
                    }
                }
                f.reset_size();

                RectF tb;
                f.typographic_bounds(gids[i], &tb);
                focr_template_t t{};
                t.letter = alphabet[i];
                t.n_w = (uint16_t)w;
                t.n_h = (uint16_t)h;
                t.offset = (uint32_t)needles.size();
                t.shift_x = (uint16_t)sx;
                t.shift_y = (uint16_t)sy;
                t.off_x = off[0];
                t.off_y = off[1];
                t.corrected_off_y = corrected[1];
                t.bearing_x = scale(tb, to_px).ox;  // src/ncc.rs:671-673
                templates.push_back(t);
                needles.insert(needles.end(), canvas.begin(), canvas.end());
            }
        }

    // advance of the first alphabet glyph (font.advance * to_px, src/ncc.rs:807)
    f.reset_size();
    float advance_px = 0.f;
    if (FT_Load_Glyph(f.face, gids[0], FT_LOAD_DEFAULT | FT_LOAD_NO_HINTING) == 0)
        advance_px = (float)f.face->glyph->metrics.horiAdvance / 64.0f * to_px;

    out->n_templates = templates.size();
    out->templates = (focr_template_t *)malloc(sizeof(focr_template_t) * (templates.size() ? templates.size() : 1));
    out->needles_len = needles.size();
    out->needles = (uint8_t *)malloc(needles.size() ? needles.size() : 1);
    if (!out->templates || !out->needles) return fail(err, errlen, "out of memory");
    memcpy(out->templates, templates.data(), sizeof(focr_template_t) * templates.size());
    memcpy(out->needles, needles.data(), needles.size());
    out->n_alphabet = (uint32_t)n_alphabet;
    out->x_bits = x_bits;
    out->y_bits = y_bits;
    out->text_size = text_size;
    out->advance_px = advance_px;
    return 0;
}

// font.metrics() as font-kit's FreeType loader reports it (src/ncc.rs:791-802 prints it under -v); from memory, unpinned
extern "C" int focr_font_metrics(const char *font_path, focr_font_metrics_t *out, char *err, size_t errlen) {
    auto fail = [&](const char *m) {
        if (err && errlen) snprintf(err, errlen, "%s", m);
        return 1;
    };
    if (!font_path || !out) return fail("focr_font_metrics: bad arguments");
    Face f;
    if (FT_Init_FreeType(&f.lib) != 0) return fail("FT_Init_FreeType failed");
    if (FT_New_Face(f.lib, font_path, 0, &f.face) != 0) return fail("cannot open font");  // Font::from_path(..).unwrap(), src/ncc.rs:792
    const FT_Face fc = f.face;
    memset(out, 0, sizeof *out);
    out->units_per_em = fc->units_per_EM;
    out->ascent = (float)fc->ascender;
    out->descent = (float)fc->descender;
    out->line_gap = (float)(fc->height + fc->descender - fc->ascender);
    out->underline_position = (float)(fc->underline_position + fc->underline_thickness / 2);
    out->underline_thickness = (float)fc->underline_thickness;
    if (const TT_OS2 *os2 = (const TT_OS2 *)FT_Get_Sfnt_Table(fc, FT_SFNT_OS2)) {
        out->cap_height = (float)os2->sCapHeight;
        out->x_height = (float)os2->sxHeight;
    }
    out->bbox[0] = (float)fc->bbox.xMin;
    out->bbox[1] = (float)fc->bbox.yMin;
    out->bbox[2] = (float)fc->bbox.xMax;
    out->bbox[3] = (float)fc->bbox.yMax;
    return 0;
}
