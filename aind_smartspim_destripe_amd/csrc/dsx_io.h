// dsx_io.h -- host side of the chunk map: chunk files <-> (pinned) staging memory on native threads.
//
// The reference keeps CO_CPUS worker processes busy decompressing / compressing chunks around the filter
// (zarr_destriper.py:1138-1172; zarr + numcodecs do the file work).  Once the filter runs at tens of
// thousands of planes per second the Python interpreter cannot even open the chunk files fast enough
// (1024 files of 2 MiB per 64-plane block), so this part is native: a static partition of the chunk list
// over std::threads, raw or zlib chunks (Blosc is not available offline), whole-file reads straight into
// the destination, writes to a temporary name + rename (a reader sees a chunk whole or not at all).
// Pure C++ (no HIP): tests/host can build it with g++.
#ifndef DSX_IO_H
#define DSX_IO_H

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace dsx {

inline void io_mkdir_parents(const std::string& file) {
  for (size_t i = 1; i < file.size(); ++i)
    if (file[i] == '/') {
      const std::string d = file.substr(0, i);
      (void)mkdir(d.c_str(), 0777);  // EEXIST is fine
    }
}

// Whole file into buf; returns bytes read, -1 if the file does not exist, -2 on another error.
inline long long io_read_file(const char* path, void* buf, size_t cap) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return errno == ENOENT ? -1 : -2;
  size_t got = 0;
  while (got < cap) {
    const ssize_t r = read(fd, (char*)buf + got, cap - got);
    if (r < 0) { if (errno == EINTR) continue; close(fd); return -2; }
    if (r == 0) break;
    got += (size_t)r;
  }
  close(fd);
  return (long long)got;
}

inline bool io_write_file_atomic(const char* path, const void* buf, size_t n) {
  const std::string tmp = std::string(path) + ".tmp";
  int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0 && errno == ENOENT) {
    io_mkdir_parents(tmp);
    fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
  }
  if (fd < 0) return false;
  size_t put = 0;
  while (put < n) {
    const ssize_t w = write(fd, (const char*)buf + put, n - put);
    if (w < 0) { if (errno == EINTR) continue; close(fd); return false; }
    put += (size_t)w;
  }
  close(fd);
  return rename(tmp.c_str(), path) == 0;
}

template <typename F>
inline std::string io_parallel(int n, int threads, F&& one) {
  std::atomic<int> next(0);
  std::mutex mu;
  std::string err;
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      const std::string e = one(i);
      if (!e.empty()) {
        std::lock_guard<std::mutex> g(mu);
        if (err.empty()) err = e;
      }
    }
  };
  threads = threads < 1 ? 1 : (threads > n ? (n > 0 ? n : 1) : threads);
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  return err;
}

// n chunk files -> dst[i] (bytes[i] each, decompressed size).  zlib_chunks: the files are zlib streams.
// A missing chunk is filled with the 16-bit fill value (zarr semantics).
inline std::string io_read_chunks(const char* const* paths, void* const* dst, const size_t* bytes, int n,
                                  int threads, bool zlib_chunks, uint16_t fill) {
  return io_parallel(n, threads, [&](int i) -> std::string {
    if (!zlib_chunks) {
      const long long got = io_read_file(paths[i], dst[i], bytes[i]);
      if (got == -1) {
        uint16_t* p = (uint16_t*)dst[i];
        for (size_t k = 0; k < bytes[i] / 2; ++k) p[k] = fill;
        return "";
      }
      if (got != (long long)bytes[i]) return std::string("short or failed read of chunk ") + paths[i];
      return "";
    }
    struct stat st;
    if (stat(paths[i], &st) != 0) {
      if (errno != ENOENT) return std::string("cannot stat chunk ") + paths[i];
      uint16_t* p = (uint16_t*)dst[i];
      for (size_t k = 0; k < bytes[i] / 2; ++k) p[k] = fill;
      return "";
    }
    std::vector<unsigned char> raw((size_t)st.st_size);
    if (io_read_file(paths[i], raw.data(), raw.size()) != (long long)raw.size())
      return std::string("short or failed read of chunk ") + paths[i];
    uLongf out_len = (uLongf)bytes[i];
    if (uncompress((Bytef*)dst[i], &out_len, raw.data(), (uLong)raw.size()) != Z_OK || out_len != bytes[i])
      return std::string("zlib: bad chunk ") + paths[i];
    return "";
  });
}

// src[i] (bytes[i]) -> chunk file paths[i]; zlib_level < 0: raw.
inline std::string io_write_chunks(const char* const* paths, const void* const* src, const size_t* bytes, int n,
                                   int threads, int zlib_level) {
  return io_parallel(n, threads, [&](int i) -> std::string {
    if (zlib_level < 0) {
      if (!io_write_file_atomic(paths[i], src[i], bytes[i])) return std::string("cannot write chunk ") + paths[i];
      return "";
    }
    uLongf cap = compressBound((uLong)bytes[i]);
    std::vector<unsigned char> z(cap);
    if (compress2(z.data(), &cap, (const Bytef*)src[i], (uLong)bytes[i], zlib_level) != Z_OK)
      return std::string("zlib: cannot compress chunk ") + paths[i];
    if (!io_write_file_atomic(paths[i], z.data(), cap)) return std::string("cannot write chunk ") + paths[i];
    return "";
  });
}

}  // namespace dsx
#endif  // DSX_IO_H
