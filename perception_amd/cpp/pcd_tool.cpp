// pcd_tool - reads a PCD file with pclhip::io::loadPCDFile (the reader the node shims use in place of
// pcl::io::loadPCDFile<pcl::PointXYZ>, icp.cpp:159 / opd.cpp:398) and writes the points as raw float32 x y z triples.
//   pcd_tool in.pcd out.bin      prints "points N width W height H"; exit code 1 when the file cannot be read
#include <cstdio>

#include "pcl_compat.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    pclhip::PointCloud<pclhip::PointXYZ> cloud;
    if (pclhip::io::loadPCDFile(argv[1], cloud) == -1) return 1;
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    for (const auto& p : cloud.points) { const float v[3] = {p.x, p.y, p.z}; std::fwrite(v, 4, 3, f); }
    std::fclose(f);
    std::printf("points %zu width %u height %u\n", cloud.size(), cloud.width, cloud.height);
    return 0;
}
