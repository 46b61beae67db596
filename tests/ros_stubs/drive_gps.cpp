// Drives the ground_plane_segmentation node shim without ROS: builds a sensor_msgs/PointCloud2 with the 32-byte
// pcl::PointXYZRGB record layout (x y z _ rgb _ _ _: what pcl_ros publishes for PointXYZRGB clouds) from a raw x,y,z,rgb
// float32 file, runs the node's main() (parameters, publishers) and its callback, and dumps what it published.
//   gps_shim_driver frame.bin out.bin [voxel_size distance_threshold]
// out.bin: int32 {point_step, width, height, n_fields, is_dense, n_coeff}, then per field {offset, datatype, count, name[16]},
// then n_coeff float32, then the data blob.
#define main node_main
#include "ros/ground_plane_segmentation_node.cpp"
#undef main
#include <cstdio>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::vector<float> raw;
    {
        FILE* f = std::fopen(argv[1], "rb");
        if (!f) return 2;
        float rec[4];
        while (std::fread(rec, 4, 4, f) == 4) raw.insert(raw.end(), rec, rec + 4);
        std::fclose(f);
    }
    const int n = (int)(raw.size() / 4);
    auto msg = std::make_shared<sensor_msgs::PointCloud2>();
    msg->header.frame_id = "camera_depth_optical_frame";
    msg->height = 480; msg->width = (uint32_t)(n / 480);
    msg->point_step = 32; msg->row_step = 32 * msg->width; msg->is_dense = false;
    const char* names[4] = {"x", "y", "z", "rgb"};
    const uint32_t offs[4] = {0, 4, 8, 16};
    for (int k = 0; k < 4; ++k) { sensor_msgs::PointField pf; pf.name = names[k]; pf.offset = offs[k]; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; msg->fields.push_back(pf); }
    msg->data.assign((size_t)n * 32, 0xAB);   // padding bytes of the INPUT are junk on purpose: they must not leak
    for (int i = 0; i < n; ++i) {
        std::memcpy(&msg->data[(size_t)i * 32], &raw[4 * (size_t)i], 12);
        std::memcpy(&msg->data[(size_t)i * 32 + 16], &raw[4 * (size_t)i + 3], 4);
    }
    // GPS_DRIVE_BAD=float64 | offset | step: a field table the device code cannot take (x,y,z must be FLOAT32 at 0/4/8 in
    // records of whole words) - the node must refuse it (ROS_ERROR, nothing published), not publish garbage
    if (const char* bad = std::getenv("GPS_DRIVE_BAD")) {
        const std::string b = bad;
        if (b == "float64") msg->fields[1].datatype = sensor_msgs::PointField::FLOAT64;
        else if (b == "offset") msg->fields[2].offset = 12;
        else if (b == "step") msg->point_step = 30;
    }
    ros::stub::set("voxel_size", argc > 3 ? std::atof(argv[3]) : 0.005);
    ros::stub::set("distance_threshold", argc > 4 ? std::atof(argv[4]) : 0.015);
    ros::stub::set("invert", true);
    node_main(argc, argv);
    callback(msg);
    const auto* out = ros::stub::last<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points");
    const auto* co = ros::stub::last<pcl_msgs::ModelCoefficients>("/ground_plane_segmentation/coefficients");
    if (!out || !co) { std::fprintf(stderr, "nothing published\n"); return 3; }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 2;
    const int32_t head[6] = {(int32_t)out->point_step, (int32_t)out->width, (int32_t)out->height, (int32_t)out->fields.size(), out->is_dense ? 1 : 0, (int32_t)co->values.size()};
    std::fwrite(head, 4, 6, f);
    for (const auto& pf : out->fields) {
        const int32_t d[3] = {(int32_t)pf.offset, (int32_t)pf.datatype, (int32_t)pf.count};
        char nm[16] = {0};
        std::snprintf(nm, sizeof(nm), "%s", pf.name.c_str());
        std::fwrite(d, 4, 3, f);
        std::fwrite(nm, 1, 16, f);
    }
    std::fwrite(co->values.data(), 4, co->values.size(), f);
    std::fwrite(out->data.data(), 1, out->data.size(), f);
    std::fclose(f);
    return 0;
}
