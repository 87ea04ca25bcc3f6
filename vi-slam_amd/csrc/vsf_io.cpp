// vsf_io.cpp -- libvsf_io.so: EuRoC ingestion in front of the hot path (include/vsf_io.h).
// Host-only C++ (g++ + zlib).  Follows the behaviour of the reference's
// vslam::dataprovider::EurocKimeraDataProvider (src/vslam/io/EurocKimeraDataProvider.cpp:21-228) and of the
// libraries it calls for this: cv::imread(IMREAD_GRAYSCALE) on grayscale PNGs (libpng transforms
// strip_16 / expand_gray_1_2_4_to_8 / strip_alpha), fast-cpp-csv-parser's CSVReader<2>, nlohmann::json field
// access, yaml-cpp node["T_BS"]["data"], Eigen's Quaterniond(Matrix3d) / normalize / toRotationMatrix.
// None of those libraries is present here: PARITY UNPINNED beyond the format specifications themselves
// (PNG / zlib / JSON are standards; the decoder is checked against an independent PNG implementation in
// tests/test_euroc_ingest.py).
#include <sys/stat.h>
#include "../../include/vsf_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <new>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  // a regular file of a sane size only: fopen opens a DIRECTORY too (an empty file name in data.csv names the
  // data directory itself), whose "size" by ftell is LONG_MAX -- an allocation that throws through the C ABI
  struct stat st;
  if (fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0 || st.st_size > (off_t)1 << 31) {
    std::fclose(f);
    return false;
  }
  const long n = (long)st.st_size;
  out.resize((size_t)n);
  const size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == (size_t)n;
}

// ---------------------------------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------------------------------
constexpr uint64_t VSF_PNG_MAX_PIXELS = 1ull << 28;

struct PngHeader {
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
};

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// Walks the chunks: IHDR first, IDAT payloads concatenated, IEND required, every CRC checked.
int png_parse(const uint8_t* d, size_t n, PngHeader& hd, std::vector<uint8_t>* idat) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!d || n < 8 || std::memcmp(d, sig, 8) != 0) return fail(VSF_IO_E_FORMAT, "png: bad signature");
  size_t pos = 8;
  bool have_ihdr = false, have_iend = false;
  while (pos + 12 <= n) {
    const uint32_t len = be32(d + pos);
    if (len > 0x7fffffffu || pos + 12 + (size_t)len > n) return fail(VSF_IO_E_FORMAT, "png: truncated chunk");
    const uint8_t* type = d + pos + 4;
    const uint8_t* data = d + pos + 8;
    const uint32_t crc = be32(data + len);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != crc) return fail(VSF_IO_E_FORMAT, "png: chunk CRC mismatch");
    if (!have_ihdr) {
      if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return fail(VSF_IO_E_FORMAT, "png: IHDR missing");
      hd.w = be32(data);
      hd.h = be32(data + 4);
      hd.depth = data[8];
      hd.ctype = data[9];
      hd.interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || hd.interlace > 1) return fail(VSF_IO_E_FORMAT, "png: bad IHDR methods");
      if (hd.w == 0 || hd.h == 0 || hd.w > 0x7fffffffu || hd.h > 0x7fffffffu) return fail(VSF_IO_E_FORMAT, "png: bad size");
      // nothing is allocated from a file-declared size beyond this bound (2^28 pixels, e.g. 16384 x 16384)
      if ((uint64_t)hd.w * hd.h > VSF_PNG_MAX_PIXELS) return fail(VSF_IO_E_FORMAT, "png: declared size exceeds 2^28 pixels");
      have_ihdr = true;
      if (!idat) return VSF_IO_OK;  // header only
    } else if (std::memcmp(type, "IDAT", 4) == 0) {
      idat->insert(idat->end(), data, data + len);
    } else if (std::memcmp(type, "IEND", 4) == 0) {
      have_iend = true;
      break;
    } else if (!(type[0] & 0x20)) {  // unknown critical chunk
      if (std::memcmp(type, "PLTE", 4) != 0) return fail(VSF_IO_E_FORMAT, "png: unknown critical chunk");
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr) return fail(VSF_IO_E_FORMAT, "png: IHDR missing");
  if (!have_iend) return fail(VSF_IO_E_FORMAT, "png: IEND missing (truncated file)");
  return VSF_IO_OK;
}

int png_check_kind(const PngHeader& hd) {
  const bool gray = hd.ctype == 0 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8 || hd.depth == 16);
  const bool gray_alpha = hd.ctype == 4 && (hd.depth == 8 || hd.depth == 16);
  const bool other_valid = (hd.ctype == 2 && (hd.depth == 8 || hd.depth == 16)) ||
                           (hd.ctype == 3 && (hd.depth == 1 || hd.depth == 2 || hd.depth == 4 || hd.depth == 8)) ||
                           (hd.ctype == 6 && (hd.depth == 8 || hd.depth == 16));
  if (!gray && !gray_alpha) {
    if (other_valid) return fail(VSF_IO_E_UNSUPPORTED, "png: colour / palette images are not decoded (grayscale only)");
    return fail(VSF_IO_E_FORMAT, "png: invalid colour type / bit depth");
  }
  if (hd.interlace) return fail(VSF_IO_E_UNSUPPORTED, "png: interlaced images are not decoded");
  return VSF_IO_OK;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

int png_decode(const uint8_t* d, size_t n, uint8_t* out, int32_t out_stride, int32_t expect_w, int32_t expect_h) {
  PngHeader hd;
  std::vector<uint8_t> idat;
  int rc = png_parse(d, n, hd, &idat);
  if (rc) return rc;
  rc = png_check_kind(hd);
  if (rc) return rc;
  if (expect_w > 0 && ((int32_t)hd.w != expect_w || (int32_t)hd.h != expect_h))
    return fail(VSF_IO_E_FORMAT, "png: image size differs from the expected frame size");
  if (!out || out_stride < (int32_t)hd.w) return fail(VSF_IO_E_INVALID, "png: bad output buffer");
  const int channels = hd.ctype == 4 ? 2 : 1;
  const size_t bits_pp = (size_t)channels * hd.depth;
  const size_t rowbytes = ((size_t)hd.w * bits_pp + 7) / 8;
  const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  // deflate expands by at most 1032:1, so a stream this short cannot hold the declared image; and zlib's
  // avail_out is 32 bits wide
  const uint64_t raw_size = ((uint64_t)rowbytes + 1) * hd.h;
  if (raw_size > (uint64_t)idat.size() * 1032 + 64 || raw_size > 0xffffffffull)
    return fail(VSF_IO_E_FORMAT, "png: compressed data is too short for the declared size");
  std::vector<uint8_t> raw((size_t)raw_size);
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) return fail(VSF_IO_E_FORMAT, "png: inflateInit failed");
  zs.next_in = idat.data();
  zs.avail_in = (uInt)idat.size();
  zs.next_out = raw.data();
  zs.avail_out = (uInt)raw.size();
  const int zr = inflate(&zs, Z_FINISH);
  const size_t produced = raw.size() - zs.avail_out;
  inflateEnd(&zs);
  if (zr != Z_STREAM_END || produced != raw.size())
    return fail(VSF_IO_E_FORMAT, "png: compressed data is corrupt or has the wrong length");
  // unfilter in place
  std::vector<uint8_t> zero(rowbytes, 0);
  for (uint32_t y = 0; y < hd.h; y++) {
    uint8_t* row = raw.data() + (size_t)y * (rowbytes + 1);
    const int ft = row[0];
    uint8_t* cur = row + 1;
    const uint8_t* up = y ? row - rowbytes : zero.data();
    switch (ft) {
      case 0:
        break;
      case 1:
        for (size_t i = bpp; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
        break;
      case 2:
        for (size_t i = 0; i < rowbytes; i++) cur[i] = (uint8_t)(cur[i] + up[i]);
        break;
      case 3:
        for (size_t i = 0; i < rowbytes; i++) {
          const int a = i >= bpp ? cur[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + ((a + up[i]) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < rowbytes; i++) {
          const int a = i >= bpp ? cur[i - bpp] : 0, c = i >= bpp ? up[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + paeth(a, up[i], c));
        }
        break;
      default:
        return fail(VSF_IO_E_FORMAT, "png: bad row filter type");
    }
    // to 8-bit gray
    uint8_t* o = out + (size_t)y * out_stride;
    if (hd.depth == 8) {
      if (channels == 1)
        std::memcpy(o, cur, hd.w);
      else
        for (uint32_t x = 0; x < hd.w; x++) o[x] = cur[2 * x];
    } else if (hd.depth == 16) {
      for (uint32_t x = 0; x < hd.w; x++) o[x] = cur[(size_t)x * 2 * channels];  // strip_16: the high byte
    } else {
      const int per = 8 / hd.depth, maxv = (1 << hd.depth) - 1, mul = 255 / maxv;
      for (uint32_t x = 0; x < hd.w; x++) {
        const int shift = 8 - hd.depth * (int)(x % per + 1);
        o[x] = (uint8_t)(((cur[x / per] >> shift) & maxv) * mul);
      }
    }
  }
  return VSF_IO_OK;
}

// ---------------------------------------------------------------------------------------------------
// CSV (fast-cpp-csv-parser CSVReader<2>, trim ' ' and '\t', no quotes, no comments; EurocKimeraDataProvider.cpp:104-117)
// ---------------------------------------------------------------------------------------------------
std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t')) a++;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t')) b--;
  return s.substr(a, b - a);
}

std::vector<std::string> split_csv(const std::string& line) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    const size_t c = line.find(',', a);
    out.push_back(trim(line.substr(a, c == std::string::npos ? std::string::npos : c - a)));
    if (c == std::string::npos) break;
    a = c + 1;
  }
  return out;
}

struct LogPack {
  uint64_t timestamp_ns;
  std::string filename;
};

int parse_csv(const std::string& path, std::vector<LogPack>& logs) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) return fail(VSF_IO_E_FILE, "cannot open " + path);
  std::vector<std::string> lines;
  {
    std::string cur;
    for (size_t i = 0; i < buf.size(); i++) {
      if (buf[i] == '\n') {
        if (!cur.empty() && cur.back() == '\r') cur.pop_back();
        lines.push_back(cur);
        cur.clear();
      } else {
        cur.push_back((char)buf[i]);
      }
    }
    if (!cur.empty()) {
      if (cur.back() == '\r') cur.pop_back();
      lines.push_back(cur);
    }
  }
  if (lines.empty()) return fail(VSF_IO_E_FORMAT, path + ": header missing");
  const std::vector<std::string> head = split_csv(lines[0]);
  int col_ts = -1, col_fn = -1;
  for (size_t i = 0; i < head.size(); i++) {
    if (head[i] == "#timestamp [ns]" && col_ts < 0)
      col_ts = (int)i;
    else if (head[i] == "filename" && col_fn < 0)
      col_fn = (int)i;
    else  // read_header(io::ignore_missing_column, ...) tolerates missing columns, not extra ones
      return fail(VSF_IO_E_FORMAT, path + ": extra column \"" + head[i] + "\" in the header");
  }
  for (size_t l = 1; l < lines.size(); l++) {
    const std::vector<std::string> f = split_csv(lines[l]);
    if (f.size() != head.size()) return fail(VSF_IO_E_FORMAT, path + ": wrong number of columns in line " + std::to_string(l + 1));
    LogPack p;
    p.timestamp_ns = 0;  // a missing column leaves the variable as it was (a fresh tuple: 0 / "")
    if (col_ts >= 0) {
      const std::string& s = f[col_ts];
      if (s.empty()) return fail(VSF_IO_E_FORMAT, path + ": empty timestamp in line " + std::to_string(l + 1));
      uint64_t v = 0;
      for (char ch : s) {
        if (ch < '0' || ch > '9') return fail(VSF_IO_E_FORMAT, path + ": bad timestamp in line " + std::to_string(l + 1));
        const uint64_t nv = v * 10 + (uint64_t)(ch - '0');
        if (nv / 10 != v) return fail(VSF_IO_E_FORMAT, path + ": timestamp overflow in line " + std::to_string(l + 1));
        v = nv;
      }
      p.timestamp_ns = v;
    }
    if (col_fn >= 0) p.filename = f[col_fn];
    logs.push_back(p);
  }
  return VSF_IO_OK;
}

// ---------------------------------------------------------------------------------------------------
// JSON (what nlohmann::json would parse; only the lookups of ParseCameraParameters are used, :119-166)
// ---------------------------------------------------------------------------------------------------
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::map<std::string, Json> obj;
  const Json* at(const std::string& k) const {
    if (kind != Object) return nullptr;
    auto it = obj.find(k);
    return it == obj.end() ? nullptr : &it->second;
  }
  const Json* at(size_t i) const { return kind == Array && i < arr.size() ? &arr[i] : nullptr; }
};

struct JsonParser {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
  }
  bool lit(const char* s) {
    const size_t n = std::strlen(s);
    if ((size_t)(end - p) >= n && std::memcmp(p, s, n) == 0) {
      p += n;
      return true;
    }
    return false;
  }
  Json value(int depth = 0) {
    Json j;
    ws();
    if (p >= end || depth > 64) {
      ok = false;
      return j;
    }
    if (*p == '{') {
      j.kind = Json::Object;
      p++;
      ws();
      if (p < end && *p == '}') {
        p++;
        return j;
      }
      while (ok) {
        ws();
        if (p >= end || *p != '"') {
          ok = false;
          break;
        }
        const std::string k = string();
        ws();
        if (!ok || p >= end || *p != ':') {
          ok = false;
          break;
        }
        p++;
        j.obj[k] = value(depth + 1);
        ws();
        if (p < end && *p == ',') {
          p++;
          continue;
        }
        if (p < end && *p == '}') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '[') {
      j.kind = Json::Array;
      p++;
      ws();
      if (p < end && *p == ']') {
        p++;
        return j;
      }
      while (ok) {
        j.arr.push_back(value(depth + 1));
        ws();
        if (p < end && *p == ',') {
          p++;
          continue;
        }
        if (p < end && *p == ']') {
          p++;
          break;
        }
        ok = false;
      }
    } else if (*p == '"') {
      j.kind = Json::String;
      j.str = string();
    } else if (lit("true")) {
      j.kind = Json::Bool;
      j.b = true;
    } else if (lit("false")) {
      j.kind = Json::Bool;
    } else if (lit("null")) {
      j.kind = Json::Null;
    } else {
      const char* q = p;
      if (q < end && *q == '-') q++;
      if (q >= end || *q < '0' || *q > '9') {
        ok = false;
        return j;
      }
      while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) q++;
      const std::string s(p, q);
      char* e = nullptr;
      j.kind = Json::Number;
      j.num = std::strtod(s.c_str(), &e);  // correctly rounded, as nlohmann's number parser (strtod) is
      if (!e || *e != 0) ok = false;
      p = q;
    }
    return j;
  }
  std::string string() {
    std::string s;
    p++;  // opening quote
    while (p < end && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= end) break;
        switch (*p) {
          case 'n': s.push_back('\n'); break;
          case 't': s.push_back('\t'); break;
          case 'r': s.push_back('\r'); break;
          case 'b': s.push_back('\b'); break;
          case 'f': s.push_back('\f'); break;
          case 'u': {  // kept as '?': the fields read here are ASCII
            if (end - p < 5) ok = false; else p += 4;
            s.push_back('?');
            break;
          }
          default: s.push_back(*p);
        }
        p++;
      } else {
        s.push_back(*p++);
      }
    }
    if (p >= end) {
      ok = false;
      return s;
    }
    p++;  // closing quote
    return s;
  }
};

int parse_calibration(const std::string& path, vsf_euroc_camera& cam) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) return fail(VSF_IO_E_FILE, "cannot open " + path);
  JsonParser jp{(const char*)buf.data(), (const char*)buf.data() + buf.size()};
  const Json root = jp.value();
  jp.ws();
  if (!jp.ok) return fail(VSF_IO_E_FORMAT, path + ": not valid JSON");
  const Json* v0 = root.at("value0");
  const Json* res = v0 ? v0->at("resolution") : nullptr;
  const Json* res0 = res ? res->at((size_t)0) : nullptr;
  const Json* intr = v0 ? v0->at("intrinsics") : nullptr;
  const Json* intr0 = intr ? intr->at((size_t)0) : nullptr;
  const Json* type = intr0 ? intr0->at("camera_type") : nullptr;
  const Json* prm = intr0 ? intr0->at("intrinsics") : nullptr;
  if (!res0 || !res0->at((size_t)0) || !res0->at(1) || res0->at((size_t)0)->kind != Json::Number ||
      res0->at(1)->kind != Json::Number || !type || type->kind != Json::String || !prm)
    return fail(VSF_IO_E_FORMAT, path + ": value0.resolution[0] / value0.intrinsics[0] missing");
  cam.width = (int32_t)res0->at((size_t)0)->num;
  cam.height = (int32_t)res0->at(1)->num;
  if (type->str != "ds")  // :159-166 "Specified camera model not implemented yet" -> null camera model
    return fail(VSF_IO_E_UNSUPPORTED, path + ": camera_type \"" + type->str + "\" (only \"ds\" is implemented, as in the reference)");
  const char* names[6] = {"fx", "fy", "cx", "cy", "xi", "alpha"};
  double* dst[6] = {&cam.fx, &cam.fy, &cam.cx, &cam.cy, &cam.xi, &cam.alpha};
  for (int i = 0; i < 6; i++) {
    const Json* f = prm->at(names[i]);
    if (!f || f->kind != Json::Number) return fail(VSF_IO_E_FORMAT, path + ": intrinsics." + names[i] + " missing");
    *dst[i] = f->num;
  }
  return VSF_IO_OK;
}

// ---------------------------------------------------------------------------------------------------
// sensor.yaml: node["T_BS"]["data"].as<std::vector<double>>() (:168-172), block mapping + flow sequence
// ---------------------------------------------------------------------------------------------------
bool parse_sensor_yaml(const std::string& path, std::vector<double>& data) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) return false;
  const std::string s((const char*)buf.data(), buf.size());
  // split into lines, drop comments
  std::vector<std::string> lines;
  size_t a = 0;
  while (a <= s.size()) {
    size_t b = s.find('\n', a);
    if (b == std::string::npos) b = s.size();
    std::string ln = s.substr(a, b - a);
    if (!ln.empty() && ln.back() == '\r') ln.pop_back();
    const size_t hash = ln.find('#');
    if (hash != std::string::npos && (hash == 0 || ln[hash - 1] == ' ' || ln[hash - 1] == '\t')) ln.erase(hash);
    lines.push_back(ln);
    a = b + 1;
  }
  auto indent_of = [](const std::string& ln) {
    size_t i = 0;
    while (i < ln.size() && ln[i] == ' ') i++;
    return i;
  };
  for (size_t i = 0; i < lines.size(); i++) {
    if (lines[i].compare(0, 5, "T_BS:") != 0) continue;
    for (size_t j = i + 1; j < lines.size(); j++) {
      const std::string& ln = lines[j];
      if (trim(ln).empty()) continue;
      const size_t ind = indent_of(ln);
      if (ind == 0) return false;  // left the T_BS block without a data key
      if (ln.compare(ind, 5, "data:") != 0) continue;
      std::string flow = ln.substr(ind + 5);
      size_t k = j;
      while (flow.find(']') == std::string::npos && ++k < lines.size()) flow += " " + lines[k];
      const size_t lb = flow.find('['), rb = flow.find(']');
      if (lb == std::string::npos || rb == std::string::npos || rb < lb) return false;
      std::string body = flow.substr(lb + 1, rb - lb - 1);
      size_t p0 = 0;
      data.clear();
      for (;;) {
        const size_t c = body.find(',', p0);
        const std::string tok = trim(body.substr(p0, c == std::string::npos ? std::string::npos : c - p0));
        if (!tok.empty()) {
          char* e = nullptr;
          const double v = std::strtod(tok.c_str(), &e);
          if (!e || *e != 0) return false;
          data.push_back(v);
        } else if (c != std::string::npos) {
          return false;
        }
        if (c == std::string::npos) break;
        p0 = c + 1;
      }
      return data.size() >= 16;
    }
    return false;
  }
  return false;
}

// Eigen::Quaterniond(Matrix3d) -> normalize() -> toRotationMatrix() (:199-202)
void renormalise_rotation(const double m[3][3], double R[9]) {
  double q[4];  // x y z w
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& v : q) v /= nrm;
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1.0 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1.0 - (txx + tyy);
}

}  // namespace

struct vsf_euroc {
  std::string root;
  std::vector<LogPack> logs;
  vsf_euroc_camera cam{};
  std::vector<uint8_t> mask;
  int32_t mask_w = 0, mask_h = 0;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  int32_t fw = 0, fh = 0;
  std::vector<uint8_t> file_buf;
};

// Nothing throws across the C ABI: std::bad_alloc / length_error from a container become an error code.
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(VSF_IO_E_FORMAT, "out of memory");
  } catch (const std::exception& e) {
    return fail(VSF_IO_E_FORMAT, std::string("exception: ") + e.what());
  } catch (...) {
    return fail(VSF_IO_E_FORMAT, "unknown exception");
  }
}

extern "C" {

const char* vsf_io_last_error(void) { return g_err.c_str(); }

int vsf_png_info(const uint8_t* file, size_t nbytes, int32_t* width, int32_t* height, int32_t* bit_depth,
                 int32_t* color_type) {
  return guarded([&]() -> int {
  PngHeader hd;
  const int rc = png_parse(file, nbytes, hd, nullptr);
  if (rc) return rc;
  if (width) *width = (int32_t)hd.w;
  if (height) *height = (int32_t)hd.h;
  if (bit_depth) *bit_depth = hd.depth;
  if (color_type) *color_type = hd.ctype;
  return VSF_IO_OK;
  });
}

int vsf_png_decode_gray8(const uint8_t* file, size_t nbytes, uint8_t* out, int32_t out_stride) {
  return guarded([&]() -> int {
  return png_decode(file, nbytes, out, out_stride, 0, 0);
  });
}

int vsf_png_read_gray8(const char* path, uint8_t* out, int32_t out_stride, int32_t expect_width,
                       int32_t expect_height) {
  return guarded([&]() -> int {
  if (!path) return fail(VSF_IO_E_INVALID, "null path");
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) return fail(VSF_IO_E_FILE, std::string("cannot open ") + path);
  return png_decode(buf.data(), buf.size(), out, out_stride, expect_width, expect_height);
  });
}

int vsf_euroc_open(const char* path_to_dataset_root, const char* path_to_calibration_file,
                   const char* path_to_mask_image, vsf_euroc** out) {
  return guarded([&]() -> int {
  if (!path_to_dataset_root || !path_to_calibration_file || !out) return fail(VSF_IO_E_INVALID, "null argument");
  *out = nullptr;
  std::unique_ptr<vsf_euroc> e(new vsf_euroc);
  e->root = path_to_dataset_root;
  int rc = parse_csv(e->root + "/mav0/cam0/data.csv", e->logs);  // :28-29
  if (rc) return rc;
  std::vector<double> d;
  if (parse_sensor_yaml(e->root + "/mav0/cam0/sensor.yaml", d)) {  // :30-32, :168-226
    // the comma initialiser of the reference repeats data[5] where data[6] belongs (:177-178)
    const double m[3][3] = {{d[0], d[1], d[2]}, {d[4], d[5], d[5]}, {d[8], d[9], d[10]}};
    renormalise_rotation(m, e->R);
    e->t[0] = d[3];
    e->t[1] = d[7];
    e->t[2] = d[11];
  }
  rc = parse_calibration(path_to_calibration_file, e->cam);  // :33-34
  if (rc) return rc;
  if (path_to_mask_image && path_to_mask_image[0]) {  // :35-40
    std::vector<uint8_t> buf;
    if (!read_file(path_to_mask_image, buf)) return fail(VSF_IO_E_FILE, std::string("cannot open ") + path_to_mask_image);
    PngHeader hd;
    rc = png_parse(buf.data(), buf.size(), hd, nullptr);
    if (rc) return rc;
    e->mask_w = (int32_t)hd.w;
    e->mask_h = (int32_t)hd.h;
    e->mask.resize((size_t)hd.w * hd.h);
    rc = png_decode(buf.data(), buf.size(), e->mask.data(), e->mask_w, 0, 0);
    if (rc) return rc;
    for (uint8_t& v : e->mask) v = v > 10 ? 255 : 0;  // cv::threshold(raw, mask, 10, 255, THRESH_BINARY)
  }
  *out = e.release();
  return VSF_IO_OK;
  });
}

void vsf_euroc_close(vsf_euroc* e) { delete e; }

int vsf_euroc_size(const vsf_euroc* e, uint64_t* n_frames) {
  return guarded([&]() -> int {
  if (!e || !n_frames) return fail(VSF_IO_E_INVALID, "null argument");
  *n_frames = e->logs.size();
  return VSF_IO_OK;
  });
}

int vsf_euroc_entry(const vsf_euroc* e, uint64_t index, uint64_t* timestamp_ns, const char** filename) {
  return guarded([&]() -> int {
  if (!e) return fail(VSF_IO_E_INVALID, "null argument");
  if (index >= e->logs.size()) return fail(VSF_IO_E_RANGE, "Reached the last line.");
  if (timestamp_ns) *timestamp_ns = e->logs[index].timestamp_ns;
  if (filename) *filename = e->logs[index].filename.c_str();
  return VSF_IO_OK;
  });
}

int vsf_euroc_frame_size(vsf_euroc* e, int32_t* width, int32_t* height) {
  return guarded([&]() -> int {
  if (!e || !width || !height) return fail(VSF_IO_E_INVALID, "null argument");
  if (e->fw == 0) {
    if (e->logs.empty()) return fail(VSF_IO_E_RANGE, "data.csv holds no frames");
    const std::string path = e->root + "/mav0/cam0/data/" + e->logs[0].filename;
    if (!read_file(path, e->file_buf)) return fail(VSF_IO_E_FILE, "cannot open " + path);
    PngHeader hd;
    const int rc = png_parse(e->file_buf.data(), e->file_buf.size(), hd, nullptr);
    if (rc) return rc;
    e->fw = (int32_t)hd.w;
    e->fh = (int32_t)hd.h;
  }
  *width = e->fw;
  *height = e->fh;
  return VSF_IO_OK;
  });
}

int vsf_euroc_frame(vsf_euroc* e, uint64_t index, double* timestamp_s, uint8_t* out, int32_t out_stride,
                    int32_t width, int32_t height) {
  return guarded([&]() -> int {
  if (!e || !out) return fail(VSF_IO_E_INVALID, "null argument");
  if (index >= e->logs.size()) return fail(VSF_IO_E_RANGE, "Reached the last line.");  // :71-74 -> std::nullopt
  const std::string path = e->root + "/mav0/cam0/data/" + e->logs[index].filename;     // :76-77
  if (!read_file(path, e->file_buf)) return fail(VSF_IO_E_FILE, "cannot open " + path);
  const int rc = png_decode(e->file_buf.data(), e->file_buf.size(), out, out_stride, width, height);
  if (rc) return rc;
  if (timestamp_s) *timestamp_s = static_cast<double>(e->logs[index].timestamp_ns) * 1e-9;  // :82
  return VSF_IO_OK;
  });
}

int vsf_euroc_camera_model(const vsf_euroc* e, vsf_euroc_camera* out) {
  return guarded([&]() -> int {
  if (!e || !out) return fail(VSF_IO_E_INVALID, "null argument");
  *out = e->cam;
  return VSF_IO_OK;
  });
}

int vsf_euroc_mask(const vsf_euroc* e, int32_t* has_mask, int32_t* width, int32_t* height, uint8_t* out,
                   int32_t out_stride) {
  return guarded([&]() -> int {
  if (!e || !has_mask) return fail(VSF_IO_E_INVALID, "null argument");
  *has_mask = e->mask.empty() ? 0 : 1;
  if (width) *width = e->mask_w;
  if (height) *height = e->mask_h;
  if (out && !e->mask.empty()) {
    if (out_stride < e->mask_w) return fail(VSF_IO_E_INVALID, "bad output stride");
    for (int32_t y = 0; y < e->mask_h; y++) std::memcpy(out + (size_t)y * out_stride, e->mask.data() + (size_t)y * e->mask_w, (size_t)e->mask_w);
  }
  return VSF_IO_OK;
  });
}

int vsf_euroc_sensor_pose(const vsf_euroc* e, double* rotation9, double* translation3) {
  return guarded([&]() -> int {
  if (!e || !rotation9 || !translation3) return fail(VSF_IO_E_INVALID, "null argument");
  std::memcpy(rotation9, e->R, sizeof(e->R));
  std::memcpy(translation3, e->t, sizeof(e->t));
  return VSF_IO_OK;
  });
}

}  // extern "C"
