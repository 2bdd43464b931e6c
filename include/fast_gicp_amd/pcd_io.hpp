// Minimal .pcd reader for the host layer (pcl::io::loadPCDFile stand-in for PointXYZ; src/align.cpp:118-125).
#pragma once
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "registration.hpp"

namespace fast_gicp {

/// returns 0 on success (like pcl::io::loadPCDFile), -1 otherwise
inline int loadPCDFile(const std::string& path, PointCloud<PointXYZ>& cloud) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs) return -1;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  size_t npoints = 0;
  std::string line, mode;
  while (std::getline(ifs, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key;
    ss >> key;
    if (key == "FIELDS") { std::string f; while (ss >> f) fields.push_back(f); }
    else if (key == "SIZE") { int v; while (ss >> v) sizes.push_back(v); }
    else if (key == "COUNT") { int v; while (ss >> v) counts.push_back(v); }
    else if (key == "POINTS") ss >> npoints;
    else if (key == "DATA") { ss >> mode; break; }
  }
  if (fields.empty() || fields.size() != sizes.size()) return -1;
  if (counts.empty()) counts.assign(fields.size(), 1);
  int off[3] = {-1, -1, -1}, col[3] = {-1, -1, -1}, stride = 0, ncol = 0;
  for (size_t i = 0; i < fields.size(); i++) {
    for (int a = 0; a < 3; a++) if (fields[i].size() == 1 && fields[i][0] == "xyz"[a]) { off[a] = stride; col[a] = ncol; }
    stride += sizes[i] * counts[i];
    ncol += counts[i];
  }
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) return -1;
  cloud.points.resize(npoints);
  if (mode == "binary") {
    std::vector<char> buf(npoints * (size_t)stride);
    ifs.read(buf.data(), (std::streamsize)buf.size());
    if ((size_t)ifs.gcount() != buf.size()) return -1;
    for (size_t i = 0; i < npoints; i++) {
      std::memcpy(&cloud.points[i].x, &buf[i * stride + off[0]], 4);
      std::memcpy(&cloud.points[i].y, &buf[i * stride + off[1]], 4);
      std::memcpy(&cloud.points[i].z, &buf[i * stride + off[2]], 4);
    }
    return 0;
  }
  if (mode == "ascii") {
    for (size_t i = 0; i < npoints && std::getline(ifs, line); i++) {
      std::istringstream ss(line);
      std::vector<double> v;
      double x;
      while (ss >> x) v.push_back(x);
      if ((int)v.size() < ncol) return -1;
      cloud.points[i].x = (float)v[col[0]]; cloud.points[i].y = (float)v[col[1]]; cloud.points[i].z = (float)v[col[2]];
    }
    return 0;
  }
  return -1;
}

}  // namespace fast_gicp
