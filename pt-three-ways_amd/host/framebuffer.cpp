#include "framebuffer.h"
#include "obj_loader.h" // IoError

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>

namespace ptw {

uint64_t Framebuffer::totalSamples() const {
  return std::accumulate(counts.begin(), counts.end(), uint64_t{0});
}

uint8_t componentToInt(double x) {
  return static_cast<uint8_t>(std::lround(std::pow(std::clamp(x, 0.0, 1.0), 1.0 / 2.2) * 255));
}

void toRgb8(int width, int height, const double *rgbSum, const uint32_t *counts, uint8_t *out) {
  const size_t n = static_cast<size_t>(width) * height;
  for (size_t i = 0; i < n; ++i) {
    // SampledPixel::result: the sum itself when no samples, else sum * (1.0 / n)
    const double scale = counts[i] == 0 ? 1.0 : 1.0 / static_cast<double>(counts[i]);
    for (int c = 0; c < 3; ++c) {
      const double mean = counts[i] == 0 ? rgbSum[i * 3 + c] : rgbSum[i * 3 + c] * scale;
      out[i * 3 + c] = componentToInt(mean);
    }
  }
}

namespace {

struct FileCloser {
  void operator()(FILE *f) const {
    if (f) std::fclose(f);
  }
};
using File = std::unique_ptr<FILE, FileCloser>;

constexpr uint32_t kRawSignature = 1;
constexpr uint32_t kRawVersion = 1;
constexpr size_t kRawPixelBytes = 3 * sizeof(double) + sizeof(uint32_t); // 28

File openOrThrow(const std::string &path, const char *mode) {
  File f(std::fopen(path.c_str(), mode));
  if (!f) throw IoError("Unable to open " + path);
  return f;
}

void readHeader(FILE *f, const std::string &path, int &width, int &height) {
  uint32_t header[4];
  if (std::fread(header, sizeof header, 1, f) != 1) throw IoError("Unable to read from " + path);
  if (header[0] != kRawSignature) throw IoError("Bad file " + path + " : bad signature");
  if (header[1] != kRawVersion) throw IoError("Bad file " + path + " : bad version");
  height = static_cast<int>(header[2]);
  width = static_cast<int>(header[3]);
}

} // namespace

void saveRaw(const std::string &path, int width, int height, const double *rgbSum,
             const uint32_t *counts) {
  File f = openOrThrow(path, "wb");
  const uint32_t header[4] = {kRawSignature, kRawVersion, static_cast<uint32_t>(height),
                              static_cast<uint32_t>(width)};
  const size_t n = static_cast<size_t>(width) * height;
  std::vector<unsigned char> bytes(sizeof header + n * kRawPixelBytes);
  std::memcpy(bytes.data(), header, sizeof header);
  unsigned char *p = bytes.data() + sizeof header;
  for (size_t i = 0; i < n; ++i, p += kRawPixelBytes) {
    std::memcpy(p, rgbSum + i * 3, 3 * sizeof(double));
    std::memcpy(p + 3 * sizeof(double), counts + i, sizeof(uint32_t));
  }
  if (std::fwrite(bytes.data(), 1, bytes.size(), f.get()) != bytes.size())
    throw IoError("Unable to write to " + path);
}

void readRawHeader(const std::string &path, int &width, int &height) {
  File f = openOrThrow(path, "rb");
  readHeader(f.get(), path, width, height);
}

void loadRawAccumulate(const std::string &path, int width, int height, double *rgbSum,
                       uint32_t *counts) {
  File f = openOrThrow(path, "rb");
  int w = 0, h = 0;
  readHeader(f.get(), path, w, h);
  if (w != width || h != height)
    throw SizeMismatch("Two differently-sized arrays were attempted to be combined");
  const size_t n = static_cast<size_t>(width) * height;
  std::vector<unsigned char> bytes(n * kRawPixelBytes);
  if (n && std::fread(bytes.data(), 1, bytes.size(), f.get()) != bytes.size())
    throw IoError("Unable to read from " + path);
  const unsigned char *p = bytes.data();
  for (size_t i = 0; i < n; ++i, p += kRawPixelBytes) {
    double rgb[3];
    uint32_t c;
    std::memcpy(rgb, p, sizeof rgb);
    std::memcpy(&c, p + sizeof rgb, sizeof c);
    rgbSum[i * 3 + 0] += rgb[0];
    rgbSum[i * 3 + 1] += rgb[1];
    rgbSum[i * 3 + 2] += rgb[2];
    counts[i] += c;
  }
}

// ---- PNG -------------------------------------------------------------------------------
namespace {

uint32_t crc32Update(uint32_t crc, const unsigned char *data, size_t len) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < len; ++i) crc = table[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}

void putBe32(std::vector<unsigned char> &v, uint32_t x) {
  v.push_back(static_cast<unsigned char>(x >> 24));
  v.push_back(static_cast<unsigned char>(x >> 16));
  v.push_back(static_cast<unsigned char>(x >> 8));
  v.push_back(static_cast<unsigned char>(x));
}

void writeChunk(FILE *f, const char type[4], const std::vector<unsigned char> &payload,
                const std::string &path) {
  std::vector<unsigned char> chunk;
  chunk.reserve(payload.size() + 12);
  putBe32(chunk, static_cast<uint32_t>(payload.size()));
  chunk.insert(chunk.end(), type, type + 4);
  chunk.insert(chunk.end(), payload.begin(), payload.end());
  uint32_t crc = crc32Update(0xffffffffu, chunk.data() + 4, chunk.size() - 4) ^ 0xffffffffu;
  putBe32(chunk, crc);
  if (std::fwrite(chunk.data(), 1, chunk.size(), f) != chunk.size())
    throw IoError("Unable to write to " + path);
}

} // namespace

void savePng(const std::string &path, int width, int height, const uint8_t *rgb8) {
  File f = openOrThrow(path, "wb");
  static const unsigned char magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  if (std::fwrite(magic, 1, 8, f.get()) != 8) throw IoError("Unable to write to " + path);

  std::vector<unsigned char> ihdr;
  putBe32(ihdr, static_cast<uint32_t>(width));
  putBe32(ihdr, static_cast<uint32_t>(height));
  ihdr.push_back(8); // bit depth
  ihdr.push_back(2); // colour type RGB
  ihdr.push_back(0); // deflate
  ihdr.push_back(0); // adaptive filtering
  ihdr.push_back(0); // no interlace
  writeChunk(f.get(), "IHDR", ihdr, path);

  // Scanlines: filter byte 0 + RGB bytes.
  const size_t rowBytes = static_cast<size_t>(width) * 3 + 1;
  std::vector<unsigned char> raw(rowBytes * height);
  for (int y = 0; y < height; ++y) {
    raw[y * rowBytes] = 0;
    std::memcpy(&raw[y * rowBytes + 1], rgb8 + static_cast<size_t>(y) * width * 3,
                static_cast<size_t>(width) * 3);
  }
  // zlib stream of stored (uncompressed) deflate blocks.
  std::vector<unsigned char> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78);
  z.push_back(0x01);
  uint32_t a = 1, b = 0; // adler32
  size_t pos = 0;
  do {
    const size_t len = std::min<size_t>(65535, raw.size() - pos);
    const bool last = pos + len == raw.size();
    z.push_back(last ? 1 : 0);
    z.push_back(static_cast<unsigned char>(len & 0xff));
    z.push_back(static_cast<unsigned char>(len >> 8));
    z.push_back(static_cast<unsigned char>(~len & 0xff));
    z.push_back(static_cast<unsigned char>((~len >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
    for (size_t i = 0; i < len; ++i) {
      a = (a + raw[pos + i]) % 65521u;
      b = (b + a) % 65521u;
    }
    pos += len;
  } while (pos < raw.size());
  putBe32(z, (b << 16) | a);
  writeChunk(f.get(), "IDAT", z, path);
  writeChunk(f.get(), "IEND", {}, path);
}

} // namespace ptw
