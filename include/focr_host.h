/*
 * focr_host.h — C ABI of the CPU-side host pieces of the `ncc` path that sit
 * either side of the device scan: template-bank rasterisation (FreeType, on
 * the CPU as BASELINE.json's north_star keeps it), bank files, image decode,
 * synthetic pages for benchmarks/tests, and the reference's output formats.
 *
 * libfocr_host.so   : everything except rasterisation (no third-party deps
 *                     beyond zlib for PNG).
 * libfocr_raster.so : focr_raster_bank only (links FreeType).
 */
#ifndef FOCR_HOST_H
#define FOCR_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "focr_ncc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct focr_bank {
    focr_template_t *templates;
    size_t n_templates;
    uint8_t *needles;
    size_t needles_len;
    uint32_t n_alphabet;          /* templates per sub-pixel offset */
    uint32_t x_bits, y_bits;      /* 2^x_bits * 2^y_bits offsets, x-major (src/ncc.rs:563-573) */
    float text_size;
    float advance_px;             /* advance of the first alphabet glyph in px (synthetic pages) */
} focr_bank_t;

enum { FOCR_BOX_FONT = 0, FOCR_BOX_ALPHABET = 1, FOCR_BOX_CHAR = 2 }; /* src/ncc.rs:33-50 */

/* Rasterise the template bank exactly as get_hits + render do for every
 * (offset, letter) (src/ncc.rs:563-573, 587-649, 143-196; box-size logic
 * 588-628).  alphabet = code points in order.  On success fills *out (free
 * with focr_bank_free).  Returns 0, or non-zero with a message in err. */
int focr_raster_bank(const char *font_path, float text_size, uint32_t x_bits, uint32_t y_bits,
                     int hinting, const uint32_t *alphabet, size_t n_alphabet, int box_size,
                     uint32_t x_padding, uint32_t y_padding, focr_bank_t *out, char *err,
                     size_t errlen);

/* font.metrics() of the reference's font-kit loader (src/ncc.rs:791-802, the `ncc -v` preamble), in font units:
 * units_per_em, ascent, descent (negative below the baseline), line_gap, underline position / thickness, cap height,
 * x height, bounding box (origin x, y, lower-right x, y).  Restated from font-kit 0.14's FreeType loader from memory
 * (ascender, descender, height + descender - ascender, OS/2 sCapHeight / sxHeight): parity unpinned.  Returns 0 or non-zero
 * with a message in err. */
typedef struct focr_font_metrics {
    uint32_t units_per_em;
    float ascent, descent, line_gap, underline_position, underline_thickness, cap_height, x_height;
    float bbox[4];
} focr_font_metrics_t;
int focr_font_metrics(const char *font_path, focr_font_metrics_t *out, char *err, size_t errlen);

void focr_bank_free(focr_bank_t *bank);
int focr_bank_save(const char *path, const focr_bank_t *bank);
int focr_bank_load(const char *path, focr_bank_t *out);

/* Image decode to 8-bit luma as image::open(..).into_luma8() (src/ncc.rs:575;
 * crate features pnm + png, Cargo.toml:10).  PGM/PPM/PBM (P4-P6, P1-P3) and
 * non-interlaced/interlaced 8/16-bit PNG.  *px is malloc'ed. */
int focr_image_load_luma8(const char *path, uint8_t **px, size_t *w, size_t *h, char *err,
                          size_t errlen);
/* Width and height from the file's header alone (no decode). */
int focr_image_probe(const char *path, size_t *w, size_t *h, char *err, size_t errlen);
/* Decode into caller memory of `cap` bytes (e.g. one slot of a page-locked batch slab, focr_host_register): binary
 * 8-bit PGM is read in place, other formats take one extra copy.  Non-zero if it does not fit or cannot be read. */
int focr_image_load_luma8_into(const char *path, uint8_t *dst, size_t cap, size_t *w, size_t *h, char *err,
                               size_t errlen);
int focr_image_save_pgm(const char *path, const uint8_t *px, size_t w, size_t h);

/* Text of one output line as `ncc` prints it (src/ncc.rs:869-876): the letters
 * of the line's characters concatenated, UTF-8, no terminator written beyond a
 * NUL when it fits.  spaces != 0 is an EXTENSION (the reference does not detect
 * spaces, README.md:46): the gap between the
 * origins of consecutive characters is filled with round(dx / advance_px) - 1
 * blanks, which is exact for a fixed-pitch font rendered with the `alphabet`
 * or `font` box.  Returns the number of bytes the text needs (may exceed cap;
 * at most cap - 1 bytes are written). */
size_t focr_line_text(const focr_hit_t *chars, size_t n, float advance_px, int spaces, char *out,
                      size_t cap);

/* Synthetic page: white background, black
 * anti-aliased text composited from the bank itself.  Left/right margin 45 px,
 * first line box at y = 39, line advance = box height, pen advance =
 * bank->advance_px; each glyph is stamped from the bank variant whose x shift
 * is the pen's fraction rounded down to the bank's grid.  Characters i.i.d.
 * uniform over the alphabet's non-blank glyphs from SplitMix64(seed).
 * luma_out: r_w*r_h bytes, 255 = paper.  truth (optional, capacity
 * truth_cap): one focr_hit_t per stamped glyph (x, y = box origin,
 * template_index = variant stamped).  Returns the number of glyphs stamped. */
size_t focr_synth_page(const focr_bank_t *bank, uint64_t seed, size_t r_w, size_t r_h,
                       uint8_t *luma_out, focr_hit_t *truth, size_t truth_cap);

/* Rust `Display` for f32 (shortest round-trip, no exponent for the magnitudes
 * that occur here), used by --csv / --raw (src/ncc.rs:685-697, 855-864).
 * Writes a NUL-terminated string, returns its length. */
size_t focr_format_f32(float v, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* FOCR_HOST_H */
