// Host-side robustness of the EuRoC ingestion (vi-slam_amd/csrc/vsf_io.cpp: PNG, csv, yaml, json readers) under
// AddressSanitizer + UBSan: built by tests/test_io_sanitizers.py together with vsf_io.cpp, run on the CPU.
//   usage: io_sanitize_main <dataset root> <calibration json> <mask png or ""> <seed> <rounds>
// 1. reads the data set as written (every frame, camera model, mask, pose) and prints a checksum;
// 2. mutates copies of a PNG (bit flips, truncations, length-field edits, random bytes) and of the three text files
//    and feeds them to the decoders / the provider: any result is fine, a memory error is not.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <zlib.h>

#include "vsf_io.h"

static uint64_t rng_state;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(rng_state >> 33);
}
static std::vector<uint8_t> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const std::vector<uint8_t>& d) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write(reinterpret_cast<const char*>(d.data()), (std::streamsize)d.size());
}
static void mutate(std::vector<uint8_t>& d) {
  if (d.empty()) return;
  switch (rnd() % 6) {
    case 0: d[rnd() % d.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
    case 1: d.resize(rnd() % (d.size() + 1)); break;
    case 2: for (int k = 0; k < 8; k++) d[rnd() % d.size()] = (uint8_t)rnd(); break;
    case 3: { size_t i = rnd() % d.size(); d.insert(d.begin() + (long)i, (size_t)(rnd() % 64), (uint8_t)rnd()); break; }
    case 4: { size_t i = rnd() % d.size(); size_t n = rnd() % 16; if (i + n > d.size()) n = d.size() - i; d.erase(d.begin() + (long)i, d.begin() + (long)(i + n)); break; }
    default: { size_t i = rnd() % d.size(); for (size_t k = i; k < d.size() && k < i + 4; k++) d[k] = 0xff; break; }
  }
}

// chunk CRCs re-made after a mutation, so that the decoder gets past its integrity checks and into the mutated
// header fields, zlib stream and filter bytes
static void fix_png_crcs(std::vector<uint8_t>& d) {
  size_t p = 8;
  while (p + 12 <= d.size()) {
    const uint32_t len = ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3];
    if ((uint64_t)p + 12 + len > d.size()) break;
    const uint32_t c = (uint32_t)crc32(0L, d.data() + p + 4, 4 + len);
    d[p + 8 + len] = (uint8_t)(c >> 24), d[p + 9 + len] = (uint8_t)(c >> 16), d[p + 10 + len] = (uint8_t)(c >> 8),
                d[p + 11 + len] = (uint8_t)c;
    p += 12 + (size_t)len;
  }
}

int main(int argc, char** argv) {
  if (argc != 6) return 2;
  const std::string root = argv[1], calib = argv[2], mask = argv[3];
  rng_state = std::strtoull(argv[4], nullptr, 10) * 2654435761ull + 12345;
  const int rounds = std::atoi(argv[5]);
  // ---- 1. the data set as written
  vsf_euroc* e = nullptr;
  if (vsf_euroc_open(root.c_str(), calib.c_str(), mask.empty() ? nullptr : mask.c_str(), &e) != 0) {
    std::fprintf(stderr, "open: %s\n", vsf_io_last_error());
    return 1;
  }
  uint64_t n = 0, sum = 0;
  int32_t w = 0, h = 0;
  vsf_euroc_size(e, &n);
  if (vsf_euroc_frame_size(e, &w, &h) != 0) return 1;
  std::vector<uint8_t> img((size_t)w * h);
  std::string first_png;
  for (uint64_t i = 0; i < n + 2; i++) {  // two past the end: must be refused, not read
    double ts = 0;
    const int rc = vsf_euroc_frame(e, i, &ts, img.data(), w, w, h);
    if (i < n && rc != 0) return 1;
    if (i >= n && rc == 0) return 1;
    if (rc == 0) for (uint8_t v : img) sum = sum * 1099511628211ull + v;
    uint64_t tns = 0;
    const char* fn = nullptr;
    if (vsf_euroc_entry(e, i, &tns, &fn) == 0 && first_png.empty()) first_png = root + "/mav0/cam0/data/" + fn;
  }
  vsf_euroc_camera cam;
  double R[9], t[3];
  int32_t hm = 0, mw = 0, mh = 0;
  if (vsf_euroc_camera_model(e, &cam) != 0 || vsf_euroc_sensor_pose(e, R, t) != 0) return 1;
  vsf_euroc_mask(e, &hm, &mw, &mh, nullptr, 0);
  if (hm) {
    std::vector<uint8_t> m((size_t)mw * mh);
    if (vsf_euroc_mask(e, &hm, &mw, &mh, m.data(), mw) != 0) return 1;
    for (uint8_t v : m) sum = sum * 1099511628211ull + v;
  }
  vsf_euroc_close(e);
  std::printf("frames %llu size %dx%d checksum %llu\n", (unsigned long long)n, w, h, (unsigned long long)sum);
  // ---- 2. mutated PNGs through the in-memory decoder
  const std::vector<uint8_t> png = slurp(first_png);
  int ok_png = 0;
  for (int r = 0; r < rounds; r++) {
    std::vector<uint8_t> d = png;
    const int k = 1 + (int)(rnd() % 3);
    for (int q = 0; q < k; q++) mutate(d);
    if (rnd() & 1) fix_png_crcs(d);
    int32_t pw = 0, ph = 0, bd = 0, ct = 0;
    if (vsf_png_info(d.data(), d.size(), &pw, &ph, &bd, &ct) == 0 && pw > 0 && ph > 0 && (long long)pw * ph <= (1 << 24)) {
      std::vector<uint8_t> out((size_t)pw * ph);
      if (vsf_png_decode_gray8(d.data(), d.size(), out.data(), pw) == 0) ok_png++;
    }
  }
  // ---- 3. mutated text files through the provider (a scratch copy of the data set's small files)
  const std::string tmp = root + "/../mut";
  const std::string mk = "mkdir -p '" + tmp + "/mav0/cam0/data'";
  if (std::system(mk.c_str()) != 0) return 1;
  spit(tmp + "/mav0/cam0/data/" + first_png.substr(first_png.rfind('/') + 1), png);
  const std::vector<uint8_t> csv = slurp(root + "/mav0/cam0/data.csv"), yaml = slurp(root + "/mav0/cam0/sensor.yaml"),
                             js = slurp(calib);
  int ok_open = 0;
  for (int r = 0; r < rounds / 4; r++) {
    std::vector<uint8_t> c = csv, y = yaml, j = js;
    const int which = (int)(rnd() % 3);
    mutate(which == 0 ? c : (which == 1 ? y : j));
    spit(tmp + "/mav0/cam0/data.csv", c);
    spit(tmp + "/mav0/cam0/sensor.yaml", y);
    spit(tmp + "/calib.json", j);
    vsf_euroc* m = nullptr;
    if (vsf_euroc_open(tmp.c_str(), (tmp + "/calib.json").c_str(), nullptr, &m) == 0) {
      ok_open++;
      uint64_t nn = 0;
      vsf_euroc_size(m, &nn);
      int32_t fw = 0, fh = 0;
      if (nn > 0 && vsf_euroc_frame_size(m, &fw, &fh) == 0 && fw > 0 && fh > 0 && (long long)fw * fh <= (1 << 24)) {
        std::vector<uint8_t> o((size_t)fw * fh);
        double ts;
        for (uint64_t i = 0; i < nn && i < 4; i++) vsf_euroc_frame(m, i, &ts, o.data(), fw, fw, fh);
      }
      vsf_euroc_camera cc;
      vsf_euroc_camera_model(m, &cc);
      vsf_euroc_sensor_pose(m, R, t);
      vsf_euroc_close(m);
    }
  }
  std::printf("mutations: %d PNGs still decoded of %d, %d data sets still opened of %d\n", ok_png, rounds, ok_open,
              rounds / 4);
  return 0;
}
