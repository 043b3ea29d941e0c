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

std::string f32s(float v);
std::string f32dbg(float v) {  // Rust `{:?}` of an f32: as `{}`, with ".0" on whole numbers
    std::string t = f32s(v);
    if (t.find_first_of(".eEn") == std::string::npos) t += ".0";  // n: inf / NaN
    return t;
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
    // a Rust panic's exit status.  _exit, not exit: once the fleet exists its worker threads may be inside HIP calls, and
    // running the HIP runtime's atexit handlers / static destructors under them can hang or crash instead of returning 101
    fflush(stdout);
    fflush(stderr);
    _exit(101);
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
    const bool timing = args.verbose || getenv("FOCR_CLI_TIMING") != nullptr;  // phase / pipeline summary lines on stderr
    PhaseClock clk{timing};
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
        // the reference's preamble, src/ncc.rs:791-802: font.metrics() and what follows from it at --text-size
        focr_font_metrics_t fm{};
        if (focr_font_metrics(args.font.c_str(), &fm, err, sizeof err) != 0) die(std::string("font metrics: ") + err);
        const float to_px = (1.f / (float)fm.units_per_em) * args.text_size;
        const float line_space = fm.ascent - fm.descent + fm.line_gap;
        fprintf(stderr, "metrics Metrics { units_per_em: %u, ascent: %s, descent: %s, line_gap: %s, underline_position: %s, underline_thickness: %s, "
                        "cap_height: %s, x_height: %s, bounding_box: RectF(<%s, %s>, <%s, %s>) }\n",
                fm.units_per_em, f32dbg(fm.ascent).c_str(), f32dbg(fm.descent).c_str(), f32dbg(fm.line_gap).c_str(), f32dbg(fm.underline_position).c_str(),
                f32dbg(fm.underline_thickness).c_str(), f32dbg(fm.cap_height).c_str(), f32dbg(fm.x_height).c_str(), f32s(fm.bbox[0]).c_str(),
                f32s(fm.bbox[1]).c_str(), f32s(fm.bbox[2]).c_str(), f32s(fm.bbox[3]).c_str());
        fprintf(stderr, "ascent  %spx\n", f32s(fm.ascent * to_px).c_str());
        fprintf(stderr, "descent %spx\n", f32s(fm.descent * to_px).c_str());
        fprintf(stderr, "font_bbox size <%s, %s>px\n", f32s(fm.bbox[2] * to_px - fm.bbox[0] * to_px).c_str(), f32s(fm.bbox[3] * to_px - fm.bbox[1] * to_px).c_str());
        fprintf(stderr, "line_space %s %spx\n", f32s(line_space).c_str(), f32s(line_space * to_px).c_str());
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

    // Pipeline (the reference's page parallelism, src/ncc.rs:839-847, on the GPU's terms):
    //   1. all host cores read the image headers, then the batch plan is fixed: consecutive pages of one size, at most
    //      kBatch per batch, each batch owning one page-aligned slab slot per page;
    //   2. the same cores decode, in index order, straight into their page's slot (binary PGM: one fread), while the
    //      main thread brings the HIP runtime up and page-locks the slabs (focr_host_register);
    //   3. a decoded batch goes to the next device round-robin (every visible GPU, FOCR_CLI_DEVICES caps the count) as
    //      ONE upload + scan + process_hits in one context of that device's executor (focr_pipe_*: FOCR_CLI_CONTEXTS lanes
    //      per device, default 3, two contexts each), queued on the device at once: a batch's DMA and small kernels run under
    //      other batches' MFMA scans and no lane waits for this thread between two batches;
    //   4. results are retired in batch order and written in page order: the bytes on stdout are those of the
    //      reference's sorted print (src/ncc.rs:845-877) whatever the device count.
    const size_t N = args.img.size();
    size_t kBatch = 128;
    if (const char *e = getenv("FOCR_CLI_BATCH")) kBatch = std::max<size_t>(1, strtoul(e, nullptr, 10));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };

    struct Page {
        size_t w = 0, h = 0, batch = 0, slot = 0;
        std::string err;
    };
    std::vector<Page> pages(N);
    unsigned n_threads = std::min(64u, std::max(1u, std::thread::hardware_concurrency()));  // more threads only fight over the address space
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // a container's CPU quota: threads beyond it only get throttled (measured: 4 096
        unsigned long long quota = 0, period = 0;           // headers in 15 ms with 16 threads on a 16-CPU quota, 130 ms with 64)
        if (fscanf(f, "%llu %llu", &quota, &period) == 2 && period) n_threads = std::min<unsigned>(n_threads, (unsigned)std::max<unsigned long long>(1, quota / period));
        fclose(f);
    }
    if (const char *e = getenv("FOCR_CLI_THREADS")) n_threads = std::max(1u, (unsigned)strtoul(e, nullptr, 10));
    n_threads = (unsigned)std::min<size_t>(n_threads, N);
    {  // 1. headers
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; t++)
            th.emplace_back([&]() {
                for (size_t i; (i = next.fetch_add(1)) < N;) {
                    char e[256] = {0};
                    if (focr_image_probe(args.img[i].c_str(), &pages[i].w, &pages[i].h, e, sizeof e) != 0) pages[i].err = e[0] ? e : "?";
                }
            });
        for (auto &t : th) t.join();
    }
    for (size_t i = 0; i < N; i++)
        if (!pages[i].err.empty()) die("cannot open image: " + pages[i].err);  // image::open(..).unwrap(), src/ncc.rs:575
    struct Batch {
        size_t p0 = 0, n = 0, w = 0, h = 0;
    };
    std::vector<Batch> batches;
    size_t max_bytes = 0;
    for (size_t i = 0; i < N; i++) {
        if (batches.empty() || batches.back().n == kBatch || batches.back().w != pages[i].w || batches.back().h != pages[i].h)
            batches.push_back(Batch{i, 0, pages[i].w, pages[i].h});
        pages[i].batch = batches.size() - 1;
        pages[i].slot = batches.back().n++;
        max_bytes = std::max(max_bytes, batches.back().n * pages[i].w * pages[i].h);
    }
    const size_t n_batches = batches.size();
    clk.lap("headers + plan");

    // devices and lanes
    size_t n_dev = 1, n_lanes = 3;
    if (const char *e = getenv("FOCR_CLI_CONTEXTS")) n_lanes = std::min<size_t>(8, std::max<size_t>(1, strtoul(e, nullptr, 10)));
    size_t dev_cap = ~(size_t)0;
    if (const char *e = getenv("FOCR_CLI_DEVICES")) dev_cap = std::max<size_t>(1, strtoul(e, nullptr, 10));
    if (args.raw) n_lanes = 1, dev_cap = 1;

    // 2. slabs: enough for every context of every device's executor to hold one batch plus a few being decoded ahead
    const size_t slab_bytes = (max_bytes + 4095) / 4096 * 4096;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<size_t> left(n_batches);
    for (size_t b = 0; b < n_batches; b++) left[b] = batches[b].n;
    std::vector<uint8_t *> slabs;          // slab of batch b = slabs[b % slabs.size()] once batch b - slabs.size() is retired
    size_t retired = 0;                    // batches whose results have been written (guarded by mu)
    size_t n_slabs = 0;                    // set once the device count is known (guarded by mu; 0 = decoders wait)
    bool stop = false;
    std::atomic<size_t> next{0};
    auto decode_worker = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < N;) {
            const size_t b = pages[i].batch;
            uint8_t *slab;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (n_slabs && b < retired + n_slabs); });
                if (stop) return;
                slab = slabs[b % n_slabs];
            }
            const size_t bytes = batches[b].w * batches[b].h;
            size_t w = 0, h = 0;
            char e[256] = {0};
            if (focr_image_load_luma8_into(args.img[i].c_str(), slab + pages[i].slot * bytes, bytes, &w, &h, e, sizeof e) != 0)
                pages[i].err = e[0] ? e : "?";
            else if (w != pages[i].w || h != pages[i].h)
                pages[i].err = args.img[i] + ": size changed between header and decode";
            std::lock_guard<std::mutex> lk(mu);
            if (--left[b] == 0) cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; t++) pool.emplace_back(decode_worker);
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
    // decoders may fill the first slabs while the HIP runtime comes up: plain page-aligned memory now, page-locked below
    {
        const size_t first = std::min<size_t>(n_batches, 4);
        std::lock_guard<std::mutex> lk(mu);
        for (size_t k = 0; k < first; k++) {
            void *p = nullptr;
            if (posix_memalign(&p, 4096, slab_bytes) != 0) die("out of memory");
            slabs.push_back((uint8_t *)p);
        }
        n_slabs = first;
    }
    cv.notify_all();

    n_dev = std::min<size_t>({(size_t)std::max(0, focr_device_count()), dev_cap, n_batches});
    if (n_dev == 0) fatal(std::string("no usable GPU: ") + focr_last_error_global());
    n_lanes = std::min(n_lanes, (n_batches + n_dev - 1) / n_dev);
    // one executor per device behind one handle (focr_fleet_*, include/focr_ncc.h): contexts and banks are set up in
    // parallel, batch b goes to device b % n_dev, fleet tickets are 1, 2, 3 ... in submission order
    focr_fleet_t *fleet = nullptr;
    {
        std::vector<int> devs(n_dev);
        for (size_t d = 0; d < n_dev; d++) devs[d] = (int)d;
        if (focr_fleet_create(devs.data(), (unsigned)n_dev, (unsigned)n_lanes, &fleet) != FOCR_OK) fatal(std::string("no usable GPU: ") + focr_last_error_global());
        if (focr_fleet_bank_upload(fleet, bank.templates, bank.n_templates, bank.needles, bank.needles_len) != FOCR_OK)
            fatal(std::string("focr_bank_upload: ") + focr_last_error_global());
        focr_fleet_set_fetch(fleet, 1);  // every lane copies its batch's counts and lines to page-locked memory itself
    }
    {  // the remaining slabs, and page-lock all of them (a slab that cannot be locked still works, through a staged copy)
        const size_t want = std::min<size_t>(n_batches, focr_fleet_slots(fleet) + 4);
        std::vector<uint8_t *> more;
        for (size_t k = slabs.size(); k < want; k++) {
            void *p = nullptr;
            if (posix_memalign(&p, 4096, slab_bytes) != 0) fatal("out of memory");
            more.push_back((uint8_t *)p);
        }
        // Growing the ring re-maps batch -> slab, so it may only happen while no decoded batch depends on the old
        // mapping: batches below `first` keep their slabs because b % first == b there and b % want == b as well.
        std::lock_guard<std::mutex> lk(mu);
        for (uint8_t *p : more) slabs.push_back(p);
        n_slabs = slabs.size();
    }
    cv.notify_all();
    for (uint8_t *p : slabs) (void)focr_host_register(p, slab_bytes);
    // --rust: the scalar scan's arithmetic and its missing cap (src/ncc.rs:320-330, 406-483) on the exact device kernel
    const int mode = args.rust ? FOCR_SCAN_RUST : FOCR_SCAN_MFMA;
    const uint32_t cap = args.rust ? 0xffffffffu : (uint32_t)FOCR_MAX_MATCHES;
    clk.lap("ctx + bank");

    const size_t T = bank.n_templates;
    double ms_wait = 0, ms_results = 0, ms_format = 0, ms_device = 0;
    std::string out;
    std::vector<uint64_t> tickets(n_batches, 0);
    std::vector<uint64_t> hits_by_letter;  // -v: hits per alphabet letter over all pages (src/ncc.rs:703-718)
    if (args.verbose) hits_by_letter.assign(alphabet.size(), 0);

    // results of batch b: read from its lane's context, formatted in page order, lane and slab released
    auto retire = [&](size_t b) {
        const Batch &B = batches[b];
        focr_host_results_t R{};
        auto t0 = now();
        if (focr_fleet_host_results(fleet, tickets[b], &R) != FOCR_OK) fatal(std::string("scan: ") + focr_last_error_global());
        const uint32_t *counts = R.counts;
        for (size_t q = 0; q < B.n * T; q++)
            if (!args.rust && counts[q] == FOCR_MAX_MATCHES) fprintf(stderr, "WARN got >= %d matches\n", FOCR_MAX_MATCHES);  // src/ncc.rs:395-397
        ms_device += R.device_ms;
        if (args.verbose) {
            // the reference's per-template line (src/ncc.rs:657-666); one device pass scans every template of every page of
            // the batch, so "elapsed" is the batch's device time divided evenly over its (page, template) pairs
            const double per_pair_ms = (double)R.device_ms / (double)(B.n * T);
            for (size_t k = 0; k < B.n; k++) {
                uint64_t page_hits = 0;
                for (size_t t = 0; t < T; t++) {
                    const focr_template_t &d = bank.templates[t];
                    const uint32_t cnt = counts[k * T + t];
                    page_hits += cnt;
                    hits_by_letter[t % alphabet.size()] += cnt;
                    fprintf(stderr, "`%s` [%s, %s] needle size %ux%u hits %u elapsed %.4fms (%.4f ns/pixel, batch average)\n",
                            utf8_encode(d.letter).c_str(), f32s(d.off_x).c_str(), f32s(d.off_y).c_str(), d.n_w, d.n_h, cnt, per_pair_ms,
                            per_pair_ms * 1e6 / (double)(B.w * B.h));
                }
                fprintf(stderr, "overall %.4fms\nhits: %llu\n", (double)R.device_ms / (double)B.n, (unsigned long long)page_hits);
            }
        }
        const uint64_t *page_off = R.page_line_off, *line_off = R.line_char_off;
        const focr_hit_t *chars = R.chars;
        ms_results += since(t0);
        t0 = now();
        for (size_t k = 0; k < B.n; k++) {  // output, src/ncc.rs:849-877
            for (uint64_t l = page_off[k]; l < page_off[k + 1]; l++) {
                const size_t nq = line_off[l + 1] - line_off[l];
                if (args.verbose) {  // process_hits' per-line histogram of the x distance between consecutive characters (src/ncc.rs:767-778)
                    std::map<int, int> dx_counts;
                    for (uint64_t q = line_off[l] + 1; q < line_off[l + 1]; q++) dx_counts[(int)chars[q].x - (int)chars[q - 1].x]++;
                    std::string h = "{";
                    for (auto &kv : dx_counts) h += (h.size() > 1 ? ", " : "") + std::to_string(kv.first) + ": " + std::to_string(kv.second);
                    fprintf(stderr, "%s}\n", h.c_str());
                }
                if (!args.csv) {
                    const size_t at = out.size();
                    out.resize(at + 4 * nq + (args.spaces ? 4096 : 0) + 1);
                    size_t need = focr_line_text(chars + line_off[l], nq, bank.advance_px, args.spaces, &out[at], out.size() - at);
                    if (need + 1 > out.size() - at) {  // a very wide gap: size exactly and redo
                        out.resize(at + need + 1);
                        need = focr_line_text(chars + line_off[l], nq, bank.advance_px, args.spaces, &out[at], out.size() - at);
                    }
                    out.resize(at + need);
                    out += '\n';
                    continue;
                }
                for (uint64_t q = line_off[l]; q < line_off[l + 1]; q++) {
                    const focr_hit_t &c = chars[q];
                    float cx = (float)c.x + (float)c.w * 0.5f, cy = (float)c.y + (float)c.h * 0.5f;
                    char row[160];
                    snprintf(row, sizeof row, "%zu,%u,%s,%s,%u,%u,%u,%u\n", B.p0 + k, c.letter, f32s(cx).c_str(), f32s(cy).c_str(), c.x, c.y, c.w,
                             c.h);
                    out += row;
                }
            }
        }
        if (focr_fleet_release(fleet, tickets[b]) != FOCR_OK) fatal("focr_fleet_release failed");  // the lane's result buffers are free again
        if (out.size() > (1u << 20) || b + 1 == n_batches) {
            fwrite(out.data(), 1, out.size(), stdout);
            out.clear();
        }
        ms_format += since(t0);
        {
            std::lock_guard<std::mutex> lk(mu);
            retired = b + 1;  // its slab may be refilled
        }
        cv.notify_all();
    };

    if (args.raw) {  // src/ncc.rs:683-698: every pre-NMS hit of the one image, in get_hits order, then exit without process_hits
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return left[0] == 0; });
        }
        if (!pages[0].err.empty()) fatal("cannot open image: " + pages[0].err);
        uint64_t ticket = 0;
        focr_ctx_t *ctx = nullptr;
        if (focr_fleet_submit(fleet, slabs[0], 0, 1, batches[0].w, batches[0].h, 1, args.threshold, cap, mode, 0, args.anchor_threshold, args.overlap, &ticket) !=
                FOCR_OK ||
            focr_fleet_wait(fleet, ticket, &ctx) != FOCR_OK)
            fatal(std::string("scan: ") + (ctx ? focr_last_error(ctx) : focr_last_error_global()));
        std::vector<uint32_t> counts(T);
        CK(ctx, focr_get_counts(ctx, counts.data()));
        for (uint32_t cnt : counts)
            if (!args.rust && cnt == FOCR_MAX_MATCHES) fprintf(stderr, "WARN got >= %d matches\n", FOCR_MAX_MATCHES);  // src/ncc.rs:395-397
        std::vector<uint64_t> off(T + 1);
        std::vector<focr_match_t> m(focr_total_matches(ctx));
        CK(ctx, focr_get_matches(ctx, off.data(), m.data()));
        for (size_t t = 0; t < T; t++) {
            const focr_template_t &d = bank.templates[t];
            for (uint64_t q = off[t]; q < off[t + 1]; q++) {
                float cx = (float)m[q].x + (float)d.n_w * 0.5f, cy = (float)m[q].y + (float)d.n_h * 0.5f;
                char row[256];
                snprintf(row, sizeof row, "%u,%s,%s,%u,%u,%u,%u,%s,%s,%s,%s\n", d.letter, f32s(cx).c_str(), f32s(cy).c_str(), m[q].x, m[q].y, d.n_w,
                         d.n_h, f32s(d.bearing_x).c_str(), f32s(d.corrected_off_y).c_str(), f32s(d.off_x).c_str(), f32s(d.off_y).c_str());
                out += row;
            }
        }
        fwrite(out.data(), 1, out.size(), stdout);
        focr_fleet_release(fleet, ticket);
        stop_pool();
        fflush(stdout);
        fflush(stderr);
        _exit(0);
    }

    // 3 + 4. submit in batch order (device b % n_dev, lanes round-robin inside the executor), retire in batch order
    const size_t max_inflight = focr_fleet_slots(fleet);  // devices x lanes x contexts per lane
    size_t next_retire = 0;
    for (size_t b = 0; b < n_batches; b++) {
        while (b - next_retire >= max_inflight) retire(next_retire++);  // the context batch b maps to still holds batch b - max_inflight
        if (b + n_dev == n_batches) (void)focr_fleet_announce_last(fleet);  // the devices' last batches: their tails need not leave room for a next scan
        auto t0 = now();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return left[b] == 0; });
        }
        ms_wait += since(t0);
        for (size_t i = batches[b].p0; i < batches[b].p0 + batches[b].n; i++)
            if (!pages[i].err.empty()) fatal("cannot open image: " + pages[i].err);  // image::open(..).unwrap(), src/ncc.rs:575
        uint8_t *slab;
        {
            std::lock_guard<std::mutex> lk(mu);
            slab = slabs[b % n_slabs];
        }
        if (focr_fleet_submit(fleet, slab, 0, batches[b].n, batches[b].w, batches[b].h, 1, args.threshold, cap, mode, 1, args.anchor_threshold, args.overlap,
                              &tickets[b]) != FOCR_OK)
            fatal(std::string("focr_fleet_submit: ") + focr_last_error_global());
    }
    while (next_retire < n_batches) retire(next_retire++);
    stop_pool();
    if (args.verbose) {
        std::vector<std::pair<uint64_t, uint32_t>> by;  // src/ncc.rs:711-718: (count, char) ascending, zero counts skipped
        for (size_t a = 0; a < alphabet.size(); a++)
            if (hits_by_letter[a]) by.push_back({hits_by_letter[a], alphabet[a]});
        std::sort(by.begin(), by.end());
        for (auto &kv : by) fprintf(stderr, "`%s` %llu\n", utf8_encode(kv.second).c_str(), (unsigned long long)kv.first);
    }
    if (timing) {
        fprintf(stderr, "pipeline: %zu batch(es) of <= %zu pages on %zu device(s) x %zu lane(s); main thread waited %.2f ms for decode; sums: device %.2f ms, "
                        "results (wait + counts + lines) %.2f ms, format %.2f ms\n",
                n_batches, kBatch, n_dev, n_lanes, ms_wait, ms_device, ms_results, ms_format);
    }
    clk.lap("pages");
    fflush(stdout);
    // Everything is written.  The executors' contexts (a dozen streams, some gigabytes of device memory) are NOT taken apart call by
    // call — 80 ms of hipFree / hipStreamDestroy in front of a process exit that reclaims all of it at once; FOCR_CLI_TEARDOWN=1 keeps
    // the orderly teardown (leak checks).
    if (getenv("FOCR_CLI_TEARDOWN")) {
        focr_fleet_destroy(fleet);
        focr_bank_free(&bank);
        clk.lap("teardown");
    }
    if (timing)
        fprintf(stderr, "total since main() %8.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - clk.t0).count());
    // Everything is flushed and the contexts are gone: leave without the HIP runtime's exit handlers (~170 ms).
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
