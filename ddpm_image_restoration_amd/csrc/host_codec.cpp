// Built-in host codec pool for the families without a device codec (SURVEY.md 8b "NULL -> built-in pool", 8f-3).
//
// Replaces the per-image Pillow calls of webp_compress / avif_compress (webp_inference.py:512-525,
// avif_inference.py:70-86) by direct calls into the SAME codec libraries through their C APIs, spread over a persistent
// C++ thread pool working on the loop's pinned staging buffers:
//   WebP  libwebp   WebPEncodeRGB(quality) -> WebPDecodeRGBInto            == Pillow's save(format="WEBP", quality=q) / open
//   AVIF  libavif   avifImageRGBToYUV (BT.709 / sRGB / BT.601, full range, 4:2:0) -> avifEncoder{quality, speed 6,
//                   autoTiling} -> avifDecoder -> avifImageYUVToRGB         == Pillow's save(format="AVIF", quality=q) / open
// Both were checked bit for bit against Pillow 12.2 in the build container (bitstream AND decoded pixels; tests/
// test_host_codec.py).  The libraries are opened at run time (dlopen): the path comes from ddpmir_set_codec_library()
// -- the Python host passes the copies bundled with its Pillow, so the results are those of the reference's own codec
// build -- then from DDPMIR_LIBWEBP / DDPMIR_LIBAVIF, then from the system's libwebp.so.7 / libavif.so.16.
// libavif's configuration structs are public but versioned; their field offsets are verified against the library's own
// defaults before first use and the codec refuses to run if they do not match (no silent garbage).
#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "host_codec.h"

namespace ddpmir {

namespace {

std::mutex g_mu;                       // guards the registry and lazy initialisation below
std::string g_path[3];                 // per DDPMIR_ARCH_*: library path given by ddpmir_set_codec_library
std::string g_err;

// ------------------------------------------------------------------------------------------------ thread pool
// Threads of this process' pool: the CPUs it may run on, shared fairly between the ranks of one host -- torch.distributed.run
// exports LOCAL_WORLD_SIZE, and with one rank per GPU every rank runs its own pool over the same cores: 8 ranks x 32 threads
// on a 128-core host would oversubscribe the codec phase that the GPUs wait for.  DDPMIR_CODEC_THREADS overrides; cap 32
// (a lane holds 8 images: more threads than images have nothing to do).
int pool_threads() {
  cpu_set_t set;
  int n = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
  int ranks = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
  n = std::max(1, n / ranks);
  if (const char* e = getenv("DDPMIR_CODEC_THREADS")) n = atoi(e);
  return std::max(1, std::min(n, 32));
}

class Pool {
 public:
  static Pool& get() { static Pool p; return p; }
  // runs fn(0..n-1) on the workers (and the caller); one batch at a time
  void parallel_for(int n, const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> batch(batch_mu_);
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = &fn; n_ = n; next_ = 0; done_ = 0;
      ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [&] { return done_ == n_; });
    fn_ = nullptr;
  }
  int size() const { return (int)threads_.size() + 1; }

 private:
  Pool() {
    const int n = pool_threads();
    for (int i = 1; i < n; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)>* f;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (!fn_ || next_ >= n_) return;
        i = next_++; f = fn_;
      }
      (*f)(i);
      {
        std::lock_guard<std::mutex> g(mu_);
        if (++done_ == n_) done_cv_.notify_all();
      }
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
      }
      work();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_, batch_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, next_ = 0, done_ = 0;
  unsigned long long epoch_ = 0;
  bool stop_ = false;
};

void* open_lib(int arch, const char* env, const char* soname, std::string& err) {
  std::vector<std::string> cands;
  if (!g_path[arch].empty()) cands.push_back(g_path[arch]);
  if (const char* e = getenv(env)) cands.push_back(e);
  cands.push_back(soname);
  for (auto& c : cands) {
    if (void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL)) return h;
    const char* why = dlerror();          // (a second call returns NULL: dlerror clears its state)
    err += c + ": " + (why ? why : "dlopen failed") + "; ";
  }
  return nullptr;
}

template <typename T>
bool sym(void* h, const char* name, T& fn, std::string& err) {
  fn = reinterpret_cast<T>(dlsym(h, name));
  if (!fn) err += std::string("missing symbol ") + name + "; ";
  return fn != nullptr;
}

// per-thread staging buffers, grown once and reused for every image (no allocation inside the sampling loop)
struct Scratch { std::vector<uint8_t> rgb, dec; };
Scratch& scratch(size_t bytes) {
  thread_local Scratch s;
  if (s.rgb.size() < bytes) { s.rgb.resize(bytes); s.dec.resize(bytes); }
  return s;
}

void planar_to_rgb(const uint8_t* chw, uint8_t* hwc, int n) {
  const uint8_t *r = chw, *g = chw + n, *b = chw + 2 * n;
  for (int i = 0; i < n; ++i) { hwc[3 * i] = r[i]; hwc[3 * i + 1] = g[i]; hwc[3 * i + 2] = b[i]; }
}
void rgb_to_planar(const uint8_t* hwc, uint8_t* chw, int n) {
  uint8_t *r = chw, *g = chw + n, *b = chw + 2 * n;
  for (int i = 0; i < n; ++i) { r[i] = hwc[3 * i]; g[i] = hwc[3 * i + 1]; b[i] = hwc[3 * i + 2]; }
}

// ------------------------------------------------------------------------------------------------ WebP
struct WebP {
  size_t (*EncodeRGB)(const uint8_t*, int, int, int, float, uint8_t**) = nullptr;
  uint8_t* (*DecodeRGBInto)(const uint8_t*, size_t, uint8_t*, size_t, int) = nullptr;
  void (*Free)(void*) = nullptr;
  int (*GetEncoderVersion)() = nullptr;
  bool ok = false, tried = false;
} g_webp;

bool webp_init() {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_webp.tried) return g_webp.ok;
  g_webp.tried = true;
  std::string err;
  void* h = open_lib(DDPMIR_ARCH_WEBP, "DDPMIR_LIBWEBP", "libwebp.so.7", err);
  if (h && sym(h, "WebPEncodeRGB", g_webp.EncodeRGB, err) && sym(h, "WebPDecodeRGBInto", g_webp.DecodeRGBInto, err) &&
      sym(h, "WebPFree", g_webp.Free, err) && sym(h, "WebPGetEncoderVersion", g_webp.GetEncoderVersion, err))
    g_webp.ok = true;
  else
    g_err = "libwebp: " + err;
  return g_webp.ok;
}

// webp_inference.py:518-523: quality = clamp(int(q), 0, 100); save(format="WEBP", quality) ; open ; RGB
int webp_image(const uint8_t* in_chw, uint8_t* out_chw, int H, int W, int quality) {
  const int n = H * W;
  Scratch& sc = scratch((size_t)3 * n);
  std::vector<uint8_t>&rgb = sc.rgb, &dec = sc.dec;
  planar_to_rgb(in_chw, rgb.data(), n);
  uint8_t* bits = nullptr;
  const size_t nb = g_webp.EncodeRGB(rgb.data(), W, H, 3 * W, (float)std::max(0, std::min(100, quality)), &bits);
  if (nb == 0 || !bits) return 1;
  uint8_t* r = g_webp.DecodeRGBInto(bits, nb, dec.data(), (size_t)3 * n, 3 * W);
  g_webp.Free(bits);
  if (!r) return 2;
  rgb_to_planar(dec.data(), out_chw, n);
  return 0;
}

// ------------------------------------------------------------------------------------------------ AVIF
// Field offsets of libavif 1.x's public structs (avif.h), verified at start-up against the library's defaults.
namespace off {
// avifImage
constexpr int im_width = 0, im_height = 4, im_depth = 8, im_yuvFormat = 12, im_yuvRange = 16, im_cicp = 104;
// avifEncoder
constexpr int en_maxThreads = 4, en_speed = 8, en_timescale = 16, en_quality = 32, en_qualityAlpha = 36, en_maxQuantizer = 44,
              en_autoTiling = 64, en_scaling = 68;
// avifRGBImage (64 bytes)
constexpr int rgb_width = 0, rgb_depth = 8, rgb_format = 12, rgb_maxThreads = 40, rgb_pixels = 48, rgb_rowBytes = 56, rgb_size = 64;
// avifDecoder
constexpr int de_maxThreads = 4, de_sizeLimit = 28, de_strict = 40, de_image = 48;
}  // namespace off

template <typename T> T rd(const void* p, int o) { T v; std::memcpy(&v, static_cast<const char*>(p) + o, sizeof v); return v; }
template <typename T> void wr(void* p, int o, T v) { std::memcpy(static_cast<char*>(p) + o, &v, sizeof v); }

struct RWData { uint8_t* data; size_t size; };

struct Avif {
  const char* (*Version)() = nullptr;
  void* (*ImageCreate)(uint32_t, uint32_t, uint32_t, int) = nullptr;
  void (*ImageDestroy)(void*) = nullptr;
  void (*RGBImageSetDefaults)(void*, const void*) = nullptr;
  int (*ImageRGBToYUV)(void*, const void*) = nullptr;
  int (*ImageYUVToRGB)(const void*, void*) = nullptr;
  void* (*EncoderCreate)() = nullptr;
  void (*EncoderDestroy)(void*) = nullptr;
  int (*EncoderAddImage)(void*, const void*, uint64_t, uint32_t) = nullptr;
  int (*EncoderFinish)(void*, RWData*) = nullptr;
  void (*RWDataFree)(RWData*) = nullptr;
  void* (*DecoderCreate)() = nullptr;
  void (*DecoderDestroy)(void*) = nullptr;
  int (*DecoderSetIOMemory)(void*, const uint8_t*, size_t) = nullptr;
  int (*DecoderParse)(void*) = nullptr;
  int (*DecoderNextImage)(void*) = nullptr;
  int enc_threads = 2;
  bool ok = false, tried = false;
} g_avif;

bool avif_layout_ok(std::string& err) {
  void* im = g_avif.ImageCreate(48, 32, 8, 3);
  void* en = g_avif.EncoderCreate();
  void* de = g_avif.DecoderCreate();
  bool ok = im && en && de;
  if (ok) {
    unsigned char rgb[off::rgb_size];
    std::memset(rgb, 0xEE, sizeof rgb);
    g_avif.RGBImageSetDefaults(rgb, im);
    ok = rd<uint32_t>(im, off::im_width) == 48 && rd<uint32_t>(im, off::im_height) == 32 && rd<uint32_t>(im, off::im_depth) == 8 &&
         rd<int>(im, off::im_yuvFormat) == 3 && rd<int>(im, off::im_yuvRange) == 1 && rd<uint16_t>(im, off::im_cicp) == 2 &&
         rd<uint16_t>(im, off::im_cicp + 2) == 2 && rd<uint16_t>(im, off::im_cicp + 4) == 2 &&
         rd<int>(en, off::en_maxThreads) == 1 && rd<int>(en, off::en_speed) == -1 && rd<uint64_t>(en, off::en_timescale) == 1 &&
         rd<int>(en, off::en_quality) == -1 && rd<int>(en, off::en_qualityAlpha) == -1 && rd<int>(en, off::en_maxQuantizer) == 63 &&
         rd<int>(en, off::en_autoTiling) == 0 && rd<int>(en, off::en_scaling) == 1 &&
         rd<uint32_t>(rgb, off::rgb_width) == 48 && rd<uint32_t>(rgb, off::rgb_depth) == 8 && rd<int>(rgb, off::rgb_format) == 1 &&
         rd<int>(rgb, off::rgb_maxThreads) == 1 && rd<uint64_t>(rgb, off::rgb_pixels) == 0 && rd<uint32_t>(rgb, off::rgb_rowBytes) == 0 &&
         rd<int>(de, off::de_maxThreads) == 1 && rd<uint32_t>(de, off::de_sizeLimit) == (1u << 28) && rd<uint32_t>(de, off::de_strict) == 7 &&
         rd<uint64_t>(de, off::de_image) == 0;     // (the decoder's image appears with avifDecoderParse)
  }
  if (!ok) err += std::string("struct layout self-check failed for libavif ") + (g_avif.Version ? g_avif.Version() : "?") + "; ";
  if (im) g_avif.ImageDestroy(im);
  if (en) g_avif.EncoderDestroy(en);
  if (de) g_avif.DecoderDestroy(de);
  return ok;
}

int avif_image(const uint8_t* in_chw, uint8_t* out_chw, int H, int W, int quality);

// Behaviour probe behind the offset table: the default-value check above cannot tell fields apart that share a default
// (quality / qualityAlpha are both -1; pixels / rowBytes / the decoder's image all 0).  So before the codec is declared usable
// one fixed 32 x 32 image goes through the very path the loop uses, at two qualities: the round trip must return the right
// geometry, quality 95 must reproduce the image closely, quality 5 clearly worse -- which only happens when the quality
// write, the RGB pixel / row-stride writes and the decoder's image pointer all land in the fields libavif reads.
bool avif_behaviour_ok(std::string& err) {
  const int H = 32, W = 32, n = H * W;
  std::vector<uint8_t> in((size_t)3 * n), hi(in.size()), lo(in.size());
  for (int c = 0; c < 3; ++c)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {   // a smooth pattern (measured with Pillow's libavif 1.4.1: mean abs error 4.4 at q95, 15.0 at q5)
        const double v = 128.0 + 60.0 * std::sin((x * (1.0 + 0.3 * c) + 0.7 * y) / 3.0) + 40.0 * std::cos((x - y) / 5.0 + c);
        in[(size_t)c * n + y * W + x] = (uint8_t)std::max(0.0, std::min(255.0, v));
      }
  const int r1 = avif_image(in.data(), hi.data(), H, W, 95), r2 = avif_image(in.data(), lo.data(), H, W, 5);
  if (r1 != 0 || r2 != 0) { err += "behaviour probe: round trip failed (" + std::to_string(r1) + ", " + std::to_string(r2) + "); "; return false; }
  double e_hi = 0.0, e_lo = 0.0;
  for (size_t i = 0; i < in.size(); ++i) { e_hi += std::abs((int)hi[i] - (int)in[i]); e_lo += std::abs((int)lo[i] - (int)in[i]); }
  e_hi /= in.size(); e_lo /= in.size();
  if (!(e_hi < 8.0 && e_lo > 1.5 * e_hi + 1.0)) {
    err += "behaviour probe: quality 95 / 5 gave mean abs errors " + std::to_string(e_hi) + " / " + std::to_string(e_lo) + " (struct fields not where expected); ";
    return false;
  }
  return true;
}

bool avif_version_ok(std::string& err) {   // the offsets were verified against libavif 1.x (avif.h of 1.0 ... 1.4)
  const char* v = g_avif.Version ? g_avif.Version() : nullptr;
  if (!v || v[0] != '1' || v[1] != '.') { err += std::string("libavif version ") + (v ? v : "?") + " is outside the verified range 1.x; "; return false; }
  return true;
}

bool avif_init() {
  std::lock_guard<std::mutex> g(g_mu);
  if (g_avif.tried) return g_avif.ok;
  g_avif.tried = true;
  std::string err;
  void* h = open_lib(DDPMIR_ARCH_AVIF, "DDPMIR_LIBAVIF", "libavif.so.16", err);
  Avif& a = g_avif;
  if (h && sym(h, "avifVersion", a.Version, err) && sym(h, "avifImageCreate", a.ImageCreate, err) &&
      sym(h, "avifImageDestroy", a.ImageDestroy, err) && sym(h, "avifRGBImageSetDefaults", a.RGBImageSetDefaults, err) &&
      sym(h, "avifImageRGBToYUV", a.ImageRGBToYUV, err) && sym(h, "avifImageYUVToRGB", a.ImageYUVToRGB, err) &&
      sym(h, "avifEncoderCreate", a.EncoderCreate, err) && sym(h, "avifEncoderDestroy", a.EncoderDestroy, err) &&
      sym(h, "avifEncoderAddImage", a.EncoderAddImage, err) && sym(h, "avifEncoderFinish", a.EncoderFinish, err) &&
      sym(h, "avifRWDataFree", a.RWDataFree, err) && sym(h, "avifDecoderCreate", a.DecoderCreate, err) &&
      sym(h, "avifDecoderDestroy", a.DecoderDestroy, err) && sym(h, "avifDecoderSetIOMemory", a.DecoderSetIOMemory, err) &&
      sym(h, "avifDecoderParse", a.DecoderParse, err) && sym(h, "avifDecoderNextImage", a.DecoderNextImage, err) &&
      avif_version_ok(err) && avif_layout_ok(err)) {
    // Pillow hands the encoder max_threads = the CPUs this process may run on (AvifImagePlugin._get_default_max_threads,
    // capped at 64 for aom).  The bitstream depends on that number only through "more than one thread" and through
    // autoTiling's tile count min(threads, ceil(area / 512^2), 32) -- checked for 1..64 threads at 128..512 pixels --
    // so the per-image encoders get Pillow's number, and avif_image() trims it to what can change the result.
    // (The count is NOT divided by LOCAL_WORLD_SIZE: it must stay Pillow's for the bitstream to be Pillow's.  What actually
    //  runs inside libavif per image is the trimmed number -- 2 threads up to 512 x 512 -- times the pool's threads above.)
    cpu_set_t set;
    int n = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("DDPMIR_AVIF_MAX_THREADS")) n = atoi(e);
    a.enc_threads = std::max(1, std::min(n, 64));
    a.ok = avif_behaviour_ok(err);
    if (!a.ok) g_err = "libavif: " + err;
  } else {
    g_err = "libavif: " + err;
  }
  return a.ok;
}

// avif_inference.py:76-84: quality = clamp(int(q), 1, 100); save(format="AVIF", quality) ; open ; RGB.
// Pillow's defaults (AvifImagePlugin._save / _avif.c): 4:2:0, full range, BT.709 primaries / sRGB transfer / BT.601
// matrix, speed 6, autotiling, timescale 1000, one frame added with AVIF_ADD_IMAGE_FLAG_SINGLE.
int avif_image(const uint8_t* in_chw, uint8_t* out_chw, int H, int W, int quality) {
  const Avif& a = g_avif;
  const int n = H * W;
  Scratch& sc = scratch((size_t)3 * n);
  std::vector<uint8_t>&rgbbuf = sc.rgb, &dec = sc.dec;
  planar_to_rgb(in_chw, rgbbuf.data(), n);
  int rc = 0;
  void *im = nullptr, *en = nullptr, *de = nullptr;
  RWData bits{nullptr, 0};
  do {
    im = a.ImageCreate((uint32_t)W, (uint32_t)H, 8, 3 /* AVIF_PIXEL_FORMAT_YUV420 */);
    if (!im) { rc = 10; break; }
    wr<int>(im, off::im_yuvRange, 1);                      // AVIF_RANGE_FULL
    wr<uint16_t>(im, off::im_cicp, 1);                     // AVIF_COLOR_PRIMARIES_BT709
    wr<uint16_t>(im, off::im_cicp + 2, 13);                // AVIF_TRANSFER_CHARACTERISTICS_SRGB
    wr<uint16_t>(im, off::im_cicp + 4, 6);                 // AVIF_MATRIX_COEFFICIENTS_BT601
    unsigned char rgb[off::rgb_size];
    a.RGBImageSetDefaults(rgb, im);
    wr<int>(rgb, off::rgb_format, 0);                      // AVIF_RGB_FORMAT_RGB
    wr<uint64_t>(rgb, off::rgb_pixels, (uint64_t)(uintptr_t)rgbbuf.data());
    wr<uint32_t>(rgb, off::rgb_rowBytes, (uint32_t)(3 * W));
    if (a.ImageRGBToYUV(im, rgb) != 0) { rc = 11; break; }
    en = a.EncoderCreate();
    if (!en) { rc = 12; break; }
    const int tiles_cap = std::min(32, (n + 512 * 512 - 1) / (512 * 512));
    wr<int>(en, off::en_maxThreads, a.enc_threads <= 1 ? 1 : std::max(2, std::min(a.enc_threads, tiles_cap)));
    wr<int>(en, off::en_speed, 6);
    wr<uint64_t>(en, off::en_timescale, 1000);
    wr<int>(en, off::en_quality, std::max(1, std::min(100, quality)));
    wr<int>(en, off::en_autoTiling, 1);
    if (a.EncoderAddImage(en, im, 1, 2 /* AVIF_ADD_IMAGE_FLAG_SINGLE */) != 0) { rc = 13; break; }
    if (a.EncoderFinish(en, &bits) != 0 || !bits.data) { rc = 14; break; }
    de = a.DecoderCreate();
    if (!de) { rc = 15; break; }
    if (a.DecoderSetIOMemory(de, bits.data, bits.size) != 0 || a.DecoderParse(de) != 0 || a.DecoderNextImage(de) != 0) { rc = 16; break; }
    const void* dim = reinterpret_cast<const void*>((uintptr_t)rd<uint64_t>(de, off::de_image));
    if (!dim || rd<uint32_t>(dim, off::im_width) != (uint32_t)W || rd<uint32_t>(dim, off::im_height) != (uint32_t)H) { rc = 17; break; }
    unsigned char rgb2[off::rgb_size];
    a.RGBImageSetDefaults(rgb2, dim);
    wr<uint32_t>(rgb2, off::rgb_depth, 8);
    wr<int>(rgb2, off::rgb_format, 0);
    wr<uint64_t>(rgb2, off::rgb_pixels, (uint64_t)(uintptr_t)dec.data());
    wr<uint32_t>(rgb2, off::rgb_rowBytes, (uint32_t)(3 * W));
    if (a.ImageYUVToRGB(dim, rgb2) != 0) { rc = 18; break; }
    rgb_to_planar(dec.data(), out_chw, n);
  } while (false);
  if (bits.data) a.RWDataFree(&bits);
  if (de) a.DecoderDestroy(de);
  if (en) a.EncoderDestroy(en);
  if (im) a.ImageDestroy(im);
  return rc;
}

// failed != null: per-image status (0 = decoded image written); the batch then succeeds as a whole and the caller supplies the
// failed images another way (the sampling loop: the reference's per-image JPEG fallback).  $DDPMIR_TEST_CODEC_FAIL=<i> makes
// image i of every batch fail without calling the codec (test hook for that fallback).
template <int (*ONE)(const uint8_t*, uint8_t*, int, int, int)>
int batch_codec_status(const uint8_t* in, uint8_t* out, int B, int H, int W, int quality, uint8_t* failed) {
  const size_t n = (size_t)3 * H * W;
  std::atomic<int> rc{0};
  int inject = -1;
  if (const char* e = getenv("DDPMIR_TEST_CODEC_FAIL")) inject = atoi(e);
  Pool::get().parallel_for(B, [&](int i) {
    int r = 1;
    if (i == inject) r = 98;
    else { try { r = ONE(in + i * n, out + i * n, H, W, quality); } catch (...) { r = 99; } }
    if (failed) failed[i] = r != 0;
    if (r != 0) rc.store(r);
  });
  return failed ? 0 : rc.load();
}
template <int (*ONE)(const uint8_t*, uint8_t*, int, int, int)>
int batch_codec(void*, const uint8_t* in, uint8_t* out, int B, int H, int W, int quality) {
  return batch_codec_status<ONE>(in, out, B, H, W, quality, nullptr);
}

}  // namespace

void set_codec_library(int arch, const std::string& path) {
  std::lock_guard<std::mutex> g(g_mu);
  if (arch < 0 || arch > 2) throw std::invalid_argument("set_codec_library: unknown codec");
  g_path[arch] = path;
  if (arch == DDPMIR_ARCH_WEBP && !g_webp.ok) g_webp.tried = false;     // a better path may succeed where the default did not
  if (arch == DDPMIR_ARCH_AVIF && !g_avif.ok) g_avif.tried = false;
}

ddpmir_codec_fn builtin_host_codec(int arch) {
  if (arch == DDPMIR_ARCH_WEBP) return webp_init() ? &batch_codec<webp_image> : nullptr;
  if (arch == DDPMIR_ARCH_AVIF) return avif_init() ? &batch_codec<avif_image> : nullptr;
  std::lock_guard<std::mutex> g(g_mu);
  g_err = "the JPEG family has a device codec; no host library is bound for it";
  return nullptr;
}

int builtin_host_codec_threads() { return pool_threads(); }

int builtin_host_codec_status(int arch, const uint8_t* in, uint8_t* out, int B, int H, int W, int quality, uint8_t* failed) {
  if (arch == DDPMIR_ARCH_WEBP && webp_init()) return batch_codec_status<webp_image>(in, out, B, H, W, quality, failed);
  if (arch == DDPMIR_ARCH_AVIF && avif_init()) return batch_codec_status<avif_image>(in, out, B, H, W, quality, failed);
  return -1;
}

std::string builtin_host_codec_error() {
  std::lock_guard<std::mutex> g(g_mu);
  return g_err;
}

}  // namespace ddpmir
