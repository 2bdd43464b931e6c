// CPU check of detail::transform_points (include/fast_gicp_amd/registration.hpp): the SSE paths for packed xyz / xyzw points must
// give the scalar loop's result bit for bit, for every tail length; other point layouts take the scalar loop. Built and run by
// tests/test_host_cpu.py.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "fast_gicp_amd/registration.hpp"
using namespace fast_gicp;
struct P16 { float x, y, z, w; };
struct P32 { float x, y, z, pad; float n[4]; };
template <typename P> int check(const char* name, size_t n) {
  std::mt19937 rng(7); std::uniform_real_distribution<float> u(-50.f, 50.f);
  std::vector<P> in(n), a(n), b(n);
  for (auto& p : in) { std::memset(&p, 0, sizeof(P)); p.x = u(rng); p.y = u(rng); p.z = u(rng); if constexpr (sizeof(P) >= 16) reinterpret_cast<float*>(&p)[3] = u(rng); }
  float m[16] = {0.99f, -0.013f, 0.021f, 0.13f, 0.012f, 0.98f, -0.033f, -2.2f, -0.02f, 0.031f, 0.97f, 30.3f, 0, 0, 0, 1};
  detail::transform_points_scalar(in.data(), a.data(), 0, n, m);
  detail::transform_points(in.data(), b.data(), n, m);
  const int bad = std::memcmp(a.data(), b.data(), n * sizeof(P)) != 0;
  double best[2] = {1e9, 1e9};
  for (int rep = 0; rep < 20; rep++) {
    auto t1 = std::chrono::high_resolution_clock::now();
    detail::transform_points_scalar(in.data(), a.data(), 0, n, m);
    auto t2 = std::chrono::high_resolution_clock::now();
    detail::transform_points(in.data(), b.data(), n, m);
    auto t3 = std::chrono::high_resolution_clock::now();
    best[0] = std::min(best[0], std::chrono::duration<double, std::micro>(t2 - t1).count());
    best[1] = std::min(best[1], std::chrono::duration<double, std::micro>(t3 - t2).count());
    if (n) m[3] += a[n / 2].x * 1e-12f;
  }
  std::printf("%s n=%zu: %s  scalar %.1f us  vector %.1f us\n", name, n, bad ? "MISMATCH" : "bit-identical", best[0], best[1]);
  return bad;
}
int main() {
  int bad = 0;
  for (size_t n : {0, 1, 3, 4, 5, 7, 17334, 100001}) { bad |= check<PointXYZ>("xyz12", n); bad |= check<P16>("xyzw16", n); }
  bad |= check<P32>("xyz+normal32", 1000);
  return bad;
}
