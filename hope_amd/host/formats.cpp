#include "formats.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace hope {
namespace {

bool slurp(const std::string& path, std::vector<uint8_t>& buf, std::string& err)
{
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open " + path; return false; }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf.resize(n > 0 ? (size_t)n : 0);
  const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  if (got != buf.size()) { err = "short read on " + path; return false; }
  return true;
}

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline int paeth(int a, int b, int c)
{
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

bool decodePng(const uint8_t* d, size_t n, Image& out, std::string& err)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (n < 8 || std::memcmp(d, sig, 8) != 0) { err = "not a PNG"; return false; }
  size_t pos = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool end = false;
  while (!end && pos + 12 <= n) {
    const uint32_t len = be32(d + pos);
    const uint8_t* tag = d + pos + 4;
    if (pos + 12 + (size_t)len > n) { err = "truncated PNG chunk"; return false; }
    const uint8_t* body = d + pos + 8;
    if (!std::memcmp(tag, "IHDR", 4)) {
      if (len < 13) { err = "bad IHDR"; return false; }
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!std::memcmp(tag, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!std::memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(tag, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || w == 0 || h == 0) { err = "PNG without IHDR"; return false; }
  if (interlace) { err = "interlaced PNG not supported"; return false; }
  int ch;
  switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: err = "bad PNG colour type"; return false;
  }
  if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) { err = "bad PNG bit depth"; return false; }
  if (w > 65535u || h > 65535u) { err = "PNG too large"; return false; }
  const size_t bits_px = (size_t)ch * depth;
  const size_t stride = (w * bits_px + 7) / 8;
  const size_t bpp = bits_px >= 8 ? bits_px / 8 : 1;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf raw_len = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || raw_len != raw.size()) { err = "PNG inflate failed"; return false; }
  // undo the scanline filters in place
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* row = raw.data() + (stride + 1) * y + 1;
    const uint8_t* up = y ? raw.data() + (stride + 1) * (y - 1) + 1 : zero.data();
    const int ft = row[-1];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
      int v = row[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: err = "bad PNG filter"; return false;
      }
      row[i] = (uint8_t)v;
    }
  }
  out = Image();
  out.width = (int)w; out.height = (int)h; out.bit_depth = depth == 16 ? 16 : 8;
  out.channels = ctype == 3 ? 3 : ch;
  const size_t npx = (size_t)w * h;
  if (depth == 16) {
    out.u16.resize(npx * ch);
    for (uint32_t y = 0; y < h; ++y) {
      const uint8_t* row = raw.data() + (stride + 1) * y + 1;
      uint16_t* o = out.u16.data() + (size_t)y * w * ch;
      for (size_t i = 0; i < (size_t)w * ch; ++i) o[i] = (uint16_t)(row[2 * i] << 8 | row[2 * i + 1]);
    }
    return true;
  }
  out.u8.resize(npx * out.channels);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* row = raw.data() + (stride + 1) * y + 1;
    uint8_t* o = out.u8.data() + (size_t)y * w * out.channels;
    for (size_t i = 0; i < (size_t)w * ch; ++i) {
      int v;
      if (depth == 8) v = row[i];
      else {
        const size_t bit = i * depth;
        v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        if (ctype != 3) v = v * 255 / ((1 << depth) - 1);
      }
      if (ctype == 3) {
        if ((size_t)v * 3 + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
        o[3 * i] = plte[3 * v]; o[3 * i + 1] = plte[3 * v + 1]; o[3 * i + 2] = plte[3 * v + 2];
      } else o[i] = (uint8_t)v;
    }
  }
  return true;
}

bool readPng(const std::string& path, Image& out, std::string& err)
{
  std::vector<uint8_t> buf;
  return slurp(path, buf, err) && decodePng(buf.data(), buf.size(), out, err);
}

bool readDepthPng(const std::string& path, std::vector<uint16_t>& depth, int& w, int& h, std::string& err)
{
  Image im;
  if (!readPng(path, im, err)) return false;
  if (im.bit_depth != 16 || im.channels != 1) { err = path + ": expected a 16-bit single-channel depth image"; return false; }
  depth.swap(im.u16); w = im.width; h = im.height;
  return true;
}

bool readTumList(const std::string& path, std::vector<TumEntry>& out, std::string& err)
{
  std::ifstream list(path.c_str());
  if (!list) { err = "Unable to open the list"; return false; }
  std::string line;
  while (std::getline(list, line)) {
    if (line.empty() || line[0] == '#') continue;
    TumEntry e;
    char rgb[512], depth[512];
    const int got = std::sscanf(line.c_str(), "%f %511s %511s %f %f %f %f %f %f %f", &e.stamp, rgb, depth, &e.t[0], &e.t[1],
                                &e.t[2], &e.q[0], &e.q[1], &e.q[2], &e.q[3]);
    if (got != 10) continue;
    e.rgb = rgb; e.depth = depth;
    out.push_back(e);
  }
  return true;
}

// ---------------------------------------------------------------------------------------- PLY / PCD
namespace {

enum Scalar { S_I8, S_U8, S_I16, S_U16, S_I32, S_U32, S_F32, S_F64, S_BAD };

int scalar_size(Scalar s) { static const int z[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return z[s]; }

Scalar ply_scalar(const std::string& t)
{
  if (t == "char" || t == "int8") return S_I8;
  if (t == "uchar" || t == "uint8") return S_U8;
  if (t == "short" || t == "int16") return S_I16;
  if (t == "ushort" || t == "uint16") return S_U16;
  if (t == "int" || t == "int32") return S_I32;
  if (t == "uint" || t == "uint32") return S_U32;
  if (t == "float" || t == "float32") return S_F32;
  if (t == "double" || t == "float64") return S_F64;
  return S_BAD;
}

double load_scalar(const uint8_t* p, Scalar s, bool swap)
{
  uint8_t b[8];
  const int n = scalar_size(s);
  for (int i = 0; i < n; ++i) b[i] = swap ? p[n - 1 - i] : p[i];
  switch (s) {
    case S_I8: return (int8_t)b[0];
    case S_U8: return b[0];
    case S_I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
    case S_U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
    case S_I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
    case S_U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
    case S_F32: { float v; std::memcpy(&v, b, 4); return v; }
    case S_F64: { double v; std::memcpy(&v, b, 8); return v; }
    default: return 0;
  }
}

struct PlyProp { std::string name; Scalar type; bool is_list; Scalar count_type; };
struct PlyElem { std::string name; size_t count; std::vector<PlyProp> props; };

bool next_line(const std::vector<uint8_t>& buf, size_t& pos, std::string& line)
{
  if (pos >= buf.size()) return false;
  size_t e = pos;
  while (e < buf.size() && buf[e] != '\n') ++e;
  line.assign((const char*)buf.data() + pos, e - pos);
  if (!line.empty() && line.back() == '\r') line.pop_back();
  pos = e + 1;
  return true;
}

// LZF as used by PCD binary_compressed (Marc Lehmann's format): literal runs and back references.
bool lzf_decompress(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out)
{
  size_t ip = 0, op = 0;
  while (ip < n_in) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      ++ctrl;
      if (op + ctrl > n_out || ip + ctrl > n_in) return false;
      std::memcpy(out + op, in + ip, ctrl);
      op += ctrl; ip += ctrl;
    } else {
      unsigned len = ctrl >> 5;
      if (ip >= n_in) return false;
      if (len == 7) { len += in[ip++]; if (ip >= n_in) return false; }
      const size_t off = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
      len += 2;
      if (off > op || op + len > n_out) return false;
      for (unsigned i = 0; i < len; ++i, ++op) out[op] = out[op - off];
    }
  }
  return op == n_out;
}

}  // namespace

bool readPly(const std::string& path, PointCloud& out, std::string& err)
{
  std::vector<uint8_t> buf;
  if (!slurp(path, buf, err)) return false;
  size_t pos = 0;
  std::string line;
  if (!next_line(buf, pos, line) || line != "ply") { err = "not a PLY file"; return false; }
  int fmt = -1;   // 0 ascii, 1 little, 2 big
  std::vector<PlyElem> elems;
  bool header_done = false;
  while (next_line(buf, pos, line)) {
    std::istringstream ss(line);
    std::string kw;
    ss >> kw;
    if (kw == "format") {
      std::string f; ss >> f;
      fmt = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
    } else if (kw == "element") {
      PlyElem e; ss >> e.name >> e.count; elems.push_back(e);
    } else if (kw == "property") {
      if (elems.empty()) { err = "PLY property before element"; return false; }
      PlyProp p; p.is_list = false; p.count_type = S_BAD;
      std::string t; ss >> t;
      if (t == "list") { std::string ct, it; ss >> ct >> it; p.is_list = true; p.count_type = ply_scalar(ct); p.type = ply_scalar(it); }
      else p.type = ply_scalar(t);
      ss >> p.name;
      if (p.type == S_BAD || (p.is_list && p.count_type == S_BAD)) { err = "PLY: unknown property type"; return false; }
      elems.back().props.push_back(p);
    } else if (kw == "end_header") { header_done = true; break; }
  }
  if (!header_done || fmt < 0) { err = "PLY: bad header"; return false; }
  const bool swap = fmt == 2;   // host is little endian
  out.clear();
  for (const PlyElem& e : elems) {
    const bool is_vertex = e.name == "vertex";
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1;
    for (size_t k = 0; k < e.props.size(); ++k) {
      const std::string& nm = e.props[k].name;
      if (nm == "x") ix = (int)k; else if (nm == "y") iy = (int)k; else if (nm == "z") iz = (int)k;
      else if (nm == "red" || nm == "diffuse_red" || nm == "r") ir = (int)k;
      else if (nm == "green" || nm == "diffuse_green" || nm == "g") ig = (int)k;
      else if (nm == "blue" || nm == "diffuse_blue" || nm == "b") ib = (int)k;
    }
    if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) { err = "PLY: vertex element without x y z"; return false; }
    if (is_vertex) out.points.reserve(e.count);
    std::vector<double> vals(e.props.size());
    for (size_t r = 0; r < e.count; ++r) {
      if (fmt == 0) {
        if (!next_line(buf, pos, line)) { err = "PLY: truncated ascii body"; return false; }
        std::istringstream ss(line);
        for (size_t k = 0; k < e.props.size(); ++k) {
          if (e.props[k].is_list) { double c = 0; ss >> c; for (int j = 0; j < (int)c; ++j) { double v; ss >> v; } vals[k] = 0; }
          else { double v = std::numeric_limits<double>::quiet_NaN(); std::string tok; ss >> tok; if (!tok.empty()) v = std::strtod(tok.c_str(), nullptr); vals[k] = v; }
        }
      } else {
        for (size_t k = 0; k < e.props.size(); ++k) {
          const PlyProp& p = e.props[k];
          if (p.is_list) {
            if (pos + scalar_size(p.count_type) > buf.size()) { err = "PLY: truncated body"; return false; }
            const size_t c = (size_t)load_scalar(buf.data() + pos, p.count_type, swap);
            pos += scalar_size(p.count_type) + c * scalar_size(p.type);
            if (pos > buf.size()) { err = "PLY: truncated body"; return false; }
            vals[k] = 0;
          } else {
            if (pos + scalar_size(p.type) > buf.size()) { err = "PLY: truncated body"; return false; }
            vals[k] = load_scalar(buf.data() + pos, p.type, swap);
            pos += scalar_size(p.type);
          }
        }
      }
      if (is_vertex) {
        PointXYZRGB p;
        p.x = (float)vals[ix]; p.y = (float)vals[iy]; p.z = (float)vals[iz];
        if (ir >= 0 && ig >= 0 && ib >= 0) { p.r = (uint8_t)vals[ir]; p.g = (uint8_t)vals[ig]; p.b = (uint8_t)vals[ib]; }
        out.points.push_back(p);
      }
    }
  }
  out.width = (uint32_t)out.points.size(); out.height = 1; out.is_dense = true;   // PLYReader leaves is_dense at true
  return true;
}

bool readPcd(const std::string& path, PointCloud& out, std::string& err)
{
  std::vector<uint8_t> buf;
  if (!slurp(path, buf, err)) return false;
  size_t pos = 0;
  std::string line;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t width = 0, height = 1, points = 0;
  int data_kind = -1;   // 0 ascii 1 binary 2 binary_compressed
  while (next_line(buf, pos, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string kw; ss >> kw;
    std::string tok;
    if (kw == "FIELDS" || kw == "COLUMNS") while (ss >> tok) fields.push_back(tok);
    else if (kw == "SIZE") while (ss >> tok) sizes.push_back(std::atoi(tok.c_str()));
    else if (kw == "TYPE") while (ss >> tok) types.push_back(tok[0]);
    else if (kw == "COUNT") while (ss >> tok) counts.push_back(std::atoi(tok.c_str()));
    else if (kw == "WIDTH") ss >> width;
    else if (kw == "HEIGHT") ss >> height;
    else if (kw == "POINTS") ss >> points;
    else if (kw == "DATA") { ss >> tok; data_kind = tok == "ascii" ? 0 : tok == "binary" ? 1 : tok == "binary_compressed" ? 2 : -1; break; }
  }
  if (data_kind < 0 || fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) { err = "PCD: bad header"; return false; }
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (points == 0) points = width * height;
  std::vector<size_t> offs(fields.size());
  size_t step = 0;
  for (size_t k = 0; k < fields.size(); ++k) { offs[k] = step; step += (size_t)sizes[k] * counts[k]; }
  int ix = -1, iy = -1, iz = -1, irgb = -1;
  for (size_t k = 0; k < fields.size(); ++k) {
    if (fields[k] == "x") ix = (int)k; else if (fields[k] == "y") iy = (int)k; else if (fields[k] == "z") iz = (int)k;
    else if (fields[k] == "rgb" || fields[k] == "rgba") irgb = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0) { err = "PCD: no x y z fields"; return false; }
  auto scalar_of = [&](int k) {
    const char t = types[k]; const int s = sizes[k];
    if (t == 'F') return s == 8 ? S_F64 : S_F32;
    if (t == 'U') return s == 1 ? S_U8 : s == 2 ? S_U16 : S_U32;
    return s == 1 ? S_I8 : s == 2 ? S_I16 : S_I32;
  };
  out.clear();
  out.points.resize(points);
  bool dense = true;
  auto set_rgb = [](PointXYZRGB& p, uint32_t v) { p.b = v & 255; p.g = (v >> 8) & 255; p.r = (v >> 16) & 255; };
  if (data_kind == 0) {
    for (size_t r = 0; r < points; ++r) {
      if (!next_line(buf, pos, line)) { err = "PCD: truncated ascii body"; return false; }
      std::istringstream ss(line);
      PointXYZRGB& p = out.points[r];
      for (size_t k = 0; k < fields.size(); ++k)
        for (int c = 0; c < counts[k]; ++c) {
          std::string tok; ss >> tok;
          if (c) continue;
          if ((int)k == ix || (int)k == iy || (int)k == iz) {
            const float v = tok == "nan" ? std::numeric_limits<float>::quiet_NaN() : (float)std::strtod(tok.c_str(), nullptr);
            ((int)k == ix ? p.x : (int)k == iy ? p.y : p.z) = v;
          } else if ((int)k == irgb) {
            uint32_t v;
            if (types[k] == 'F') { const float f = (float)std::strtod(tok.c_str(), nullptr); std::memcpy(&v, &f, 4); }
            else v = (uint32_t)std::strtoul(tok.c_str(), nullptr, 10);
            set_rgb(p, v);
          }
        }
    }
  } else {
    std::vector<uint8_t> body;
    const uint8_t* base;
    const bool soa = data_kind == 2;
    if (data_kind == 1) {
      if (pos + step * points > buf.size()) { err = "PCD: truncated binary body"; return false; }
      base = buf.data() + pos;
    } else {
      if (pos + 8 > buf.size()) { err = "PCD: truncated compressed body"; return false; }
      uint32_t csz, usz;
      std::memcpy(&csz, buf.data() + pos, 4); std::memcpy(&usz, buf.data() + pos + 4, 4);
      if (pos + 8 + csz > buf.size() || usz != step * points) { err = "PCD: bad compressed sizes"; return false; }
      body.resize(usz);
      if (!lzf_decompress(buf.data() + pos + 8, csz, body.data(), usz)) { err = "PCD: LZF stream corrupt"; return false; }
      base = body.data();
    }
    // binary: array of records; binary_compressed: one array per field, fields in header order
    std::vector<size_t> col_base(fields.size());
    size_t acc = 0;
    for (size_t k = 0; k < fields.size(); ++k) { col_base[k] = acc; acc += (size_t)sizes[k] * counts[k] * points; }
    auto at = [&](int k, size_t r) { return soa ? base + col_base[k] + r * (size_t)sizes[k] * counts[k] : base + r * step + offs[k]; };
    for (size_t r = 0; r < points; ++r) {
      PointXYZRGB& p = out.points[r];
      p.x = (float)load_scalar(at(ix, r), scalar_of(ix), false);
      p.y = (float)load_scalar(at(iy, r), scalar_of(iy), false);
      p.z = (float)load_scalar(at(iz, r), scalar_of(iz), false);
      if (irgb >= 0) { uint32_t v; std::memcpy(&v, at(irgb, r), 4); set_rgb(p, v); }
    }
  }
  for (const PointXYZRGB& p : out.points)
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) { dense = false; break; }
  out.width = (uint32_t)width; out.height = (uint32_t)height; out.is_dense = dense;
  if ((size_t)out.width * out.height != points) { out.width = (uint32_t)points; out.height = 1; }
  return true;
}

bool writePcdBinary(const std::string& path, const PointCloud& cloud, std::string& err)
{
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) { err = "cannot write " + path; return false; }
  std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
                  "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n", cloud.size(), cloud.size());
  for (const PointXYZRGB& p : cloud.points) {
    const uint32_t rgb = (uint32_t)p.r << 16 | (uint32_t)p.g << 8 | p.b;
    float rec[4] = {p.x, p.y, p.z, 0.f};
    std::memcpy(&rec[3], &rgb, 4);
    std::fwrite(rec, 4, 4, f);
  }
  std::fclose(f);
  return true;
}

bool fromPointCloud2(const uint8_t* data, size_t n_bytes, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step,
                     const std::vector<PointField>& fields, bool is_bigendian, bool is_dense, PointCloudMono& out, std::string& err)
{
  const PointField* fx = nullptr; const PointField* fy = nullptr; const PointField* fz = nullptr;
  for (const PointField& f : fields) { if (f.name == "x") fx = &f; else if (f.name == "y") fy = &f; else if (f.name == "z") fz = &f; }
  if (!fx || !fy || !fz) { err = "PointCloud2 without x y z fields"; return false; }
  for (const PointField* f : {fx, fy, fz})
    if ((f->datatype != 7 && f->datatype != 8) || f->offset + (f->datatype == 8 ? 8u : 4u) > point_step) { err = "PointCloud2: x y z must be FLOAT32/FLOAT64 inside point_step"; return false; }
  if (row_step < width * point_step || (size_t)row_step * height > n_bytes) { err = "PointCloud2: data shorter than row_step*height"; return false; }
  out.points.resize((size_t)width * height);
  out.width = width; out.height = height; out.is_dense = is_dense;
  for (uint32_t r = 0; r < height; ++r)
    for (uint32_t c = 0; c < width; ++c) {
      const uint8_t* p = data + (size_t)r * row_step + (size_t)c * point_step;
      PointXYZ& q = out.points[(size_t)r * width + c];
      q.x = (float)load_scalar(p + fx->offset, fx->datatype == 8 ? S_F64 : S_F32, is_bigendian);
      q.y = (float)load_scalar(p + fy->offset, fy->datatype == 8 ? S_F64 : S_F32, is_bigendian);
      q.z = (float)load_scalar(p + fz->offset, fz->datatype == 8 ? S_F64 : S_F32, is_bigendian);
    }
  return true;
}

}  // namespace hope
