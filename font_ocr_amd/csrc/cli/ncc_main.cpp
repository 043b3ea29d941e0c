// ncc — the reference's `ncc` command line (src/ncc.rs:486-542, 788-878) on top of the MI355X scan.
//
// Same flags, defaults and output formats as the reference binary (clap derive struct `Args`,
// src/ncc.rs:486-542).  Differences, all deliberate:
//   * the template bank is rasterised once per run, not once per page (src/ncc.rs:561, 587-639);
//   * pages of equal size are scanned as one device batch instead of one rayon task per page
//     (src/ncc.rs:839-847); output order is still the order of -i;
//   * --rust runs the exact v_dot4 device kernel with the arithmetic and skips of the reference's scalar Rust
//     scan and without the 1024 cap (FOCR_SCAN_RUST; src/ncc.rs:320-330, 406-483);
//   * a page without any hit prints nothing (the reference panics in partition_by, src/ncc.rs:1040).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "focr_host.h"

namespace {

const char *DEFAULT_ALPHABET = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789=+<>(){};:/-";  // src/ncc.rs:28-29

struct Args {
    std::vector<std::string> img;
    std::string font;
    float text_size = 0.f;
    bool have_text_size = false;
    uint32_t x_bits = 0, y_bits = 0;
    bool hinting = false;
    float threshold = 0.8f, anchor_threshold = 0.95f;
    int overlap = 5;
    std::string alphabet = DEFAULT_ALPHABET;
    std::string box_size = "alphabet";
    uint32_t x_padding = 0, y_padding = 0;
    bool save_letters = false, rust = false, verbose = false, csv = false, raw = false;
    bool allow_wide = false;  // extension: templates 17..32 px wide instead of the reference's panic (src/ncc.rs:392)
    bool spaces = false;  // extension: fill gaps between characters with blanks (the reference does not, README.md:46)
};

[[noreturn]] void usage_error(const std::string &msg) {
    fprintf(stderr, "error: %s\n\nUsage: ncc [OPTIONS] --font <FONT> --text-size <TEXT_SIZE>\n\nFor more information, try '--help'.\n", msg.c_str());
    exit(2);  // clap's usage-error exit code
}

void print_help() {
    puts("Usage: ncc [OPTIONS] --font <FONT> --text-size <TEXT_SIZE>\n\nOptions:\n"
         "  -i, --img <IMG>...                         \n"
         "  -f, --font <FONT>                          \n"
         "  -t, --text-size <TEXT_SIZE>                \n"
         "      --x-bits <X_BITS>                      [default: 0]\n"
         "      --y-bits <Y_BITS>                      [default: 0]\n"
         "      --hinting                              \n"
         "      --threshold <THRESHOLD>                [default: 0.8]\n"
         "      --anchor-threshold <ANCHOR_THRESHOLD>  [default: 0.95]\n"
         "      --overlap <OVERLAP>                    [default: 5]\n"
         "  -a, --alphabet <ALPHABET>                  [default: ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789=+<>(){};:/-]\n"
         "      --box-size <BOX_SIZE>                  [default: alphabet]\n"
         "      --x-padding <X_PADDING>                [default: 0]\n"
         "      --y-padding <Y_PADDING>                [default: 0]\n"
         "      --save-letters                         \n"
         "      --rust                                 \n"
         "  -v, --verbose                              \n"
         "      --csv                                  \n"
         "      --raw                                  \n"
         "      --spaces                               [extension] print blanks for gaps of whole advances\n"
         "      --allow-wide                           [extension] accept templates 17..32 px wide\n"
         "  -h, --help                                 Print help\n"
         "  -V, --version                              Print version");
}

Args parse_args(int argc, char **argv) {
    Args a;
    std::vector<std::string> v(argv + 1, argv + argc);
    auto is_opt = [](const std::string &s) { return s.size() > 1 && s[0] == '-' && !(isdigit((unsigned char)s[1]) || s[1] == '.'); };
    for (size_t i = 0; i < v.size(); i++) {
        std::string k = v[i], val;
        bool has_val = false;
        if (k.rfind("--", 0) == 0) {
            size_t eq = k.find('=');
            if (eq != std::string::npos) {
                val = k.substr(eq + 1);
                k = k.substr(0, eq);
                has_val = true;
            }
        } else if (k.size() > 2 && k[0] == '-' && k[1] != '-') {  // -t13, -ipage.pgm
            val = k.substr(k[2] == '=' ? 3 : 2);
            k = k.substr(0, 2);
            has_val = true;
        }
        auto need = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= v.size()) usage_error("a value is required for '" + k + "' but none was supplied");
            return v[++i];
        };
        auto num_u = [&](const std::string &s) -> uint32_t {
            char *e = nullptr;
            unsigned long r = strtoul(s.c_str(), &e, 10);
            if (!e || *e || s.empty() || s[0] == '-') usage_error("invalid value '" + s + "' for '" + k + "'");
            return (uint32_t)r;
        };
        auto num_f = [&](const std::string &s) -> float {
            char *e = nullptr;
            float r = strtof(s.c_str(), &e);
            if (!e || *e || s.empty()) usage_error("invalid value '" + s + "' for '" + k + "'");
            return r;
        };
        if (k == "-i" || k == "--img") {
            a.img.push_back(need());
            while (i + 1 < v.size() && !is_opt(v[i + 1])) a.img.push_back(v[++i]);  // num_args = 1..
        } else if (k == "-f" || k == "--font") a.font = need();
        else if (k == "-t" || k == "--text-size") a.text_size = num_f(need()), a.have_text_size = true;
        else if (k == "--x-bits") a.x_bits = num_u(need());
        else if (k == "--y-bits") a.y_bits = num_u(need());
        else if (k == "--hinting") a.hinting = true;
        else if (k == "--threshold") a.threshold = num_f(need());
        else if (k == "--anchor-threshold") a.anchor_threshold = num_f(need());
        else if (k == "--overlap") {
            std::string s = need();
            char *e = nullptr;
            long r = strtol(s.c_str(), &e, 10);
            if (!e || *e || s.empty()) usage_error("invalid value '" + s + "' for '--overlap <OVERLAP>'");
            a.overlap = (int)r;
        } else if (k == "-a" || k == "--alphabet") a.alphabet = need();
        else if (k == "--box-size") a.box_size = need();
        else if (k == "--x-padding") a.x_padding = num_u(need());
        else if (k == "--y-padding") a.y_padding = num_u(need());
        else if (k == "--save-letters") a.save_letters = true;
        else if (k == "--rust") a.rust = true;
        else if (k == "-v" || k == "--verbose") a.verbose = true;
        else if (k == "--csv") a.csv = true;
        else if (k == "--raw") a.raw = true;
        else if (k == "--spaces") a.spaces = true;
        else if (k == "--allow-wide") a.allow_wide = true;
        else if (k == "-h" || k == "--help") {
            print_help();
            exit(0);
        } else if (k == "-V" || k == "--version") {
            puts("ncc 0.1.0");
            exit(0);
        } else usage_error("unexpected argument '" + v[i] + "' found");
    }
    if (a.font.empty()) usage_error("the following required arguments were not provided:\n  --font <FONT>");
    if (!a.have_text_size) usage_error("the following required arguments were not provided:\n  --text-size <TEXT_SIZE>");
    return a;
}

std::vector<uint32_t> utf8_decode(const std::string &s) {
    std::vector<uint32_t> out;
    for (size_t i = 0; i < s.size();) {
        unsigned char c = (unsigned char)s[i];
        uint32_t cp;
        int n;
        if (c < 0x80) cp = c, n = 1;
        else if ((c >> 5) == 6) cp = c & 0x1f, n = 2;
        else if ((c >> 4) == 14) cp = c & 0x0f, n = 3;
        else cp = c & 0x07, n = 4;
        for (int k = 1; k < n && i + k < s.size(); k++) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3f);
        out.push_back(cp);
        i += n;
    }
    return out;
}

std::string utf8_encode(uint32_t cp) {
    std::string s;
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) s += (char)(0xc0 | (cp >> 6)), s += (char)(0x80 | (cp & 0x3f));
    else if (cp < 0x10000) s += (char)(0xe0 | (cp >> 12)), s += (char)(0x80 | ((cp >> 6) & 0x3f)), s += (char)(0x80 | (cp & 0x3f));
    else s += (char)(0xf0 | (cp >> 18)), s += (char)(0x80 | ((cp >> 12) & 0x3f)), s += (char)(0x80 | ((cp >> 6) & 0x3f)), s += (char)(0x80 | (cp & 0x3f));
    return s;
}

std::string f32s(float v) {  // Rust `{}` of an f32
    char buf[64];
    focr_format_f32(v, buf, sizeof buf);
    return buf;
}

// --save-letters: letters/{letter}-{x}_{y}.png, 8-bit grey (src/ncc.rs:642-649)
bool write_png_gray(const std::string &path, const uint8_t *px, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw((size_t)(w + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        raw[(size_t)y * (w + 1)] = 0;
        memcpy(&raw[(size_t)y * (w + 1) + 1], px + (size_t)y * w, w);
    }
    uLongf zl = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zl);
    if (compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; };
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t hdr[8];
        be32(hdr, len);
        memcpy(hdr + 4, type, 4);
        fwrite(hdr, 1, 8, f);
        if (len) fwrite(data, 1, len, f);
        uLong crc = crc32(0, (const Bytef *)type, 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t c[4];
        be32(c, (uint32_t)crc);
        fwrite(c, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {0};
    be32(ihdr, w);
    be32(ihdr + 4, h);
    ihdr[8] = 8;  // bit depth, colour type 0 (grey)
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zl);
    chunk("IEND", nullptr, 0);
    fclose(f);
    return true;
}

[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "ncc: %s\n", msg.c_str());
    exit(101);  // a Rust panic's exit status
}

#define CK(ctx, expr)                                                        \
    do {                                                                     \
        if ((expr) != FOCR_OK) die(std::string(#expr) + ": " + focr_last_error(ctx)); \
    } while (0)

}  // namespace

struct PhaseClock {  // -v: wall time of each host phase, on stderr
    bool on;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), t = t0;
    void lap(const char *what) {
        auto n = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "phase %-14s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

int main(int argc, char **argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // more streams than the default 4 hardware queues must not share one
    Args args = parse_args(argc, argv);
    PhaseClock clk{args.verbose};
    int box = args.box_size == "font" ? FOCR_BOX_FONT : args.box_size == "alphabet" ? FOCR_BOX_ALPHABET : args.box_size == "char" ? FOCR_BOX_CHAR : -1;
    if (box < 0) die("called `Result::unwrap()` on an `Err` value: () (--box-size must be font, alphabet or char)");  // src/ncc.rs:559
    if (args.raw && args.img.size() != 1) die("assertion failed: args.img.len() == 1");  // src/ncc.rs:834

    // template bank, once (get_hits re-does this per page: src/ncc.rs:561, 587-639)
    std::vector<uint32_t> alphabet = utf8_decode(args.alphabet);
    focr_bank_t bank{};
    char err[256] = {0};
    if (focr_raster_bank(args.font.c_str(), args.text_size, args.x_bits, args.y_bits, args.hinting, alphabet.data(), alphabet.size(),
                         box, args.x_padding, args.y_padding, &bank, err, sizeof err) != 0)
        die(std::string("rasterising the template bank failed: ") + err);
    clk.lap("bank raster");
    if (args.verbose) {
        fprintf(stderr, "bank: %zu templates (%zu letters x %u x %u sub-pixel offsets), advance %spx\n", bank.n_templates,
                alphabet.size(), 1u << args.x_bits, 1u << args.y_bits, f32s(bank.advance_px).c_str());
    }
    for (size_t t = 0; t < bank.n_templates; t++)
        if (bank.templates[t].n_w > 16 && !args.allow_wide) die("not handled");  // src/ncc.rs:392
    if (args.save_letters) {
        mkdir("letters", 0777);
        for (size_t t = 0; t < bank.n_templates; t++) {
            const focr_template_t &d = bank.templates[t];
            std::string path = "letters/" + utf8_encode(d.letter) + "-" + std::to_string((size_t)(d.off_x * 1000.f)) + "_" +
                               std::to_string((size_t)(d.off_y * 1000.f)) + ".png";
            if (!write_png_gray(path, bank.needles + d.offset, d.n_w, d.n_h)) die("cannot write " + path);
        }
    }
    if (args.img.empty()) return 0;

    // Pipeline: images are decoded by all host cores in index order, in batches of
    // kBatch pages; the GPU context comes up meanwhile; the main thread scans batch b while batches b+1.. decode.
    // Pages of one batch are grouped by size (one resident device batch per size).  Output is produced per page
    // and written in page order, so the bytes on stdout are those of the reference's sorted print (src/ncc.rs:845-877).
    const size_t N = args.img.size();
    size_t kBatch = 256, kAhead = 5;  // decode-ahead in batches; must exceed the contexts in flight (<= 4)
    if (const char *e = getenv("FOCR_CLI_BATCH")) kBatch = std::max<size_t>(1, strtoul(e, nullptr, 10));
    struct Page {
        uint8_t *px = nullptr;
        size_t w = 0, h = 0;
        std::string err;
    };
    std::vector<Page> pages(N);
    const size_t n_batches = (N + kBatch - 1) / kBatch;
    std::vector<size_t> left(n_batches);
    for (size_t b = 0; b < n_batches; b++) left[b] = std::min(kBatch, N - b * kBatch);
    std::mutex mu;
    std::condition_variable cv;
    size_t consumed = 0;  // batches the main thread has finished with (guarded by mu)
    bool stop = false;
    std::atomic<size_t> next{0};
    auto decode_worker = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < N;) {
            const size_t b = i / kBatch;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || b < consumed + kAhead; });  // bound the decoded pages held in memory
                if (stop) return;
            }
            char e[256] = {0};
            if (focr_image_load_luma8(args.img[i].c_str(), &pages[i].px, &pages[i].w, &pages[i].h, e, sizeof e) != 0)
                pages[i].err = e[0] ? e : "?";
            std::lock_guard<std::mutex> lk(mu);
            if (--left[b] == 0) cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    {
        unsigned hw = std::min(64u, std::thread::hardware_concurrency());  // more threads only fight over the address space
        if (const char *e = getenv("FOCR_CLI_THREADS")) hw = (unsigned)strtoul(e, nullptr, 10);
        unsigned nt = std::max(1u, std::min<unsigned>(hw, (unsigned)std::min<size_t>(N, kBatch * kAhead)));
        for (unsigned t = 0; t < nt; t++) pool.emplace_back(decode_worker);
    }
    auto stop_pool = [&]() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : pool) t.join();
        pool.clear();
    };
    auto fatal = [&](const std::string &msg) {  // panic with the decoders stopped first
        stop_pool();
        die(msg);
    };

    // FOCR_CLI_CONTEXTS=n (default 1): n device contexts take the batches alternately, each batch on its own host
    // thread, so that one batch's upload / statistics / sort / verify overlap another's MFMA scan (the scheme of
    // bench.py --in-flight 2).  Measured on image files it does not pay: this loop is bound by the pageable
    // host<->device copies, not by the scan, hence the default.
    size_t n_ctx = 1;
    if (const char *e = getenv("FOCR_CLI_CONTEXTS")) n_ctx = std::min<size_t>(4, std::max<size_t>(1, strtoul(e, nullptr, 10)));
    n_ctx = std::min(n_ctx, n_batches);
    if (args.raw) n_ctx = 1;
    std::vector<focr_ctx_t *> ctxs(n_ctx, nullptr);
    for (size_t j = 0; j < n_ctx; j++) {
        if (focr_ctx_create(0, &ctxs[j]) != FOCR_OK) fatal(std::string("no usable GPU: ") + focr_last_error_global());
        if (focr_bank_upload(ctxs[j], bank.templates, bank.n_templates, bank.needles, bank.needles_len) != FOCR_OK)
            fatal(std::string("focr_bank_upload: ") + focr_last_error(ctxs[j]));
    }
    // --rust: the scalar scan's arithmetic and its missing cap (src/ncc.rs:320-330, 406-483) on the exact device kernel
    const int mode = args.rust ? FOCR_SCAN_RUST : FOCR_SCAN_MFMA;
    const uint32_t cap = args.rust ? 0xffffffffu : (uint32_t)FOCR_MAX_MATCHES;
    clk.lap("ctx + bank");

    const size_t T = bank.n_templates;
    struct BatchOut {
        std::string text, err, log;
        double ms_upload = 0, ms_scan = 0, ms_post = 0, ms_format = 0;
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };

    // Everything the device does for one decoded batch; runs on a worker thread, touches only its own context.
    auto process_batch = [&](size_t b, focr_ctx_t *ctx) -> BatchOut {
        BatchOut r;
        const size_t p0 = b * kBatch, p1 = std::min(N, p0 + kBatch);
#define CKB(expr)                                                        \
    do {                                                                 \
        if ((expr) != FOCR_OK) {                                         \
            r.err = std::string(#expr) + ": " + focr_last_error(ctx);    \
            return r;                                                    \
        }                                                                \
    } while (0)
        std::map<std::pair<size_t, size_t>, std::vector<size_t>> groups;
        for (size_t i = p0; i < p1; i++) {
            if (!pages[i].err.empty()) {  // image::open(..).unwrap(), src/ncc.rs:575
                r.err = "cannot open image: " + pages[i].err;
                return r;
            }
            groups[{pages[i].w, pages[i].h}].push_back(i);
        }
        std::vector<std::string> page_out(p1 - p0);
        for (auto &kv : groups) {
            const size_t w = kv.first.first, h = kv.first.second;
            const std::vector<size_t> &idx = kv.second;
            auto t0 = now();
            CKB(focr_pages_alloc(ctx, idx.size(), w, h));
            for (size_t k = 0; k < idx.size(); k++) CKB(focr_pages_upload(ctx, k, 1, pages[idx[k]].px, 1));
            r.ms_upload += since(t0);
            t0 = now();
            CKB(focr_scan(ctx, args.threshold, cap, mode));
            r.ms_scan += since(t0);
            t0 = now();
            std::vector<uint32_t> counts(idx.size() * T);
            CKB(focr_get_counts(ctx, counts.data()));
            for (uint32_t cnt : counts)
                if (!args.rust && cnt == FOCR_MAX_MATCHES) r.log += "WARN got >= " + std::to_string(FOCR_MAX_MATCHES) + " matches\n";  // src/ncc.rs:395-397
            if (args.verbose) {
                float ms[6];
                focr_last_timings(ctx, ms);
                char line[160];
                snprintf(line, sizeof line, "scan of %zu page(s) %zux%zu: %.3fms on the device, hits: %zu\n", idx.size(), w, h, ms[5],
                         focr_total_matches(ctx));
                r.log += line;
            }
            if (args.raw) {  // src/ncc.rs:683-698: every pre-NMS hit, in get_hits order (exactly one image: src/ncc.rs:834)
                std::vector<uint64_t> off(idx.size() * T + 1);
                std::vector<focr_match_t> m(focr_total_matches(ctx));
                CKB(focr_get_matches(ctx, off.data(), m.data()));
                for (size_t t = 0; t < T; t++) {
                    const focr_template_t &d = bank.templates[t];
                    for (uint64_t q = off[t]; q < off[t + 1]; q++) {
                        float cx = (float)m[q].x + (float)d.n_w * 0.5f, cy = (float)m[q].y + (float)d.n_h * 0.5f;
                        char row[256];
                        snprintf(row, sizeof row, "%u,%s,%s,%u,%u,%u,%u,%s,%s,%s,%s\n", d.letter, f32s(cx).c_str(), f32s(cy).c_str(), m[q].x,
                                 m[q].y, d.n_w, d.n_h, f32s(d.bearing_x).c_str(), f32s(d.corrected_off_y).c_str(), f32s(d.off_x).c_str(),
                                 f32s(d.off_y).c_str());
                        r.text += row;
                    }
                }
                return r;
            }
            CKB(focr_process_hits(ctx, args.anchor_threshold, args.overlap));
            std::vector<uint64_t> page_off(idx.size() + 1), line_off(focr_total_lines(ctx) + 1);
            std::vector<focr_hit_t> chars(focr_total_chars(ctx));
            CKB(focr_get_lines(ctx, page_off.data(), line_off.data(), chars.data()));
            r.ms_post += since(t0);
            t0 = now();
            for (size_t k = 0; k < idx.size(); k++) {  // output, src/ncc.rs:849-877
                std::string &s = page_out[idx[k] - p0];
                for (uint64_t l = page_off[k]; l < page_off[k + 1]; l++) {
                    if (!args.csv) {
                        const size_t nq = line_off[l + 1] - line_off[l];
                        std::string line(4 * nq + (args.spaces ? 4096 : 0) + 1, '\0');
                        size_t need = focr_line_text(chars.data() + line_off[l], nq, bank.advance_px, args.spaces, &line[0], line.size());
                        if (need + 1 > line.size()) {  // a very wide gap: size exactly and redo
                            line.assign(need + 1, '\0');
                            need = focr_line_text(chars.data() + line_off[l], nq, bank.advance_px, args.spaces, &line[0], line.size());
                        }
                        line.resize(need);
                        s += line;
                        s += '\n';
                        continue;
                    }
                    for (uint64_t q = line_off[l]; q < line_off[l + 1]; q++) {
                        const focr_hit_t &c = chars[q];
                        float cx = (float)c.x + (float)c.w * 0.5f, cy = (float)c.y + (float)c.h * 0.5f;
                        char row[160];
                        snprintf(row, sizeof row, "%zu,%u,%s,%s,%u,%u,%u,%u\n", idx[k], c.letter, f32s(cx).c_str(), f32s(cy).c_str(), c.x,
                                 c.y, c.w, c.h);
                        s += row;
                    }
                }
            }
            r.ms_format += since(t0);
        }
#undef CKB
        for (size_t i = p0; i < p1; i++) {
            r.text += page_out[i - p0];
            free(pages[i].px);
            pages[i].px = nullptr;
        }
        return r;
    };

    double ms_wait = 0, ms_upload = 0, ms_scan = 0, ms_post = 0, ms_format = 0;
    std::deque<std::future<BatchOut>> inflight;  // oldest first: results are written in batch order
    std::string out;
    size_t retired = 0;
    auto retire = [&]() {
        BatchOut r = inflight.front().get();
        inflight.pop_front();
        fputs(r.log.c_str(), stderr);
        if (!r.err.empty()) {
            for (auto &f : inflight) f.wait();
            fatal(r.err);
        }
        ms_upload += r.ms_upload, ms_scan += r.ms_scan, ms_post += r.ms_post, ms_format += r.ms_format;
        out += r.text;
        retired++;
        if (out.size() > (1u << 20) || retired == n_batches) {
            fwrite(out.data(), 1, out.size(), stdout);
            out.clear();
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            consumed = retired;
        }
        cv.notify_all();
    };
    for (size_t b = 0; b < n_batches; b++) {
        if (inflight.size() == n_ctx) retire();  // context b % n_ctx is the one that ran batch b - n_ctx
        auto t0 = now();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return left[b] == 0; });  // b <= retired + n_ctx - 1 < consumed + kAhead: decoders may reach it
        }
        ms_wait += since(t0);
        inflight.push_back(std::async(std::launch::async, process_batch, b, ctxs[b % n_ctx]));
    }
    while (!inflight.empty()) retire();
    stop_pool();
    if (args.verbose)
        fprintf(stderr, "pipeline: %zu batch(es) of <= %zu pages on %zu context(s); main thread waited %.2f ms for decode; per-batch sums: "
                        "upload %.2f ms, scan %.2f ms, counts+process_hits+fetch %.2f ms, format %.2f ms\n",
                n_batches, kBatch, n_ctx, ms_wait, ms_upload, ms_scan, ms_post, ms_format);
    clk.lap("pages");
    fflush(stdout);
    for (focr_ctx_t *c : ctxs) focr_ctx_destroy(c);
    focr_bank_free(&bank);
    clk.lap("teardown");
    if (args.verbose)
        fprintf(stderr, "total since main() %8.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clk.t0).count());
    // Everything is flushed and the context is gone: leave without the HIP runtime's exit handlers (~170 ms).
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
