// Drives the ground_plane_segmentation and iterative_closest_point node shims back to back without ROS, the way
// iterative_closest_point.launch wires them (gps -> /ground_plane_segmentation/points -> icp): a raw x,y,z,rgb float32
// frame goes into the gps callback as a 32-byte-record PointCloud2, what gps published goes into the icp callback, and
// what icp published (/icp/pose, /icp/bbox_points, /icp/aligned_points, /icp/template, TF) is dumped as text.
//   chain_shim_driver frame.bin template.pcd [icp_fitness_score]
// The two nodes are separate translation units (gps_node.o / icp_node.o, main renamed); this file only sees their entry points.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <geometry_msgs/Pose.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

int gps_node_main(int argc, char** argv);
int icp_node_main(int argc, char** argv);
void callback(const sensor_msgs::PointCloud2ConstPtr& input);          // ground_plane_segmentation_node.cpp
void icp_callback(const sensor_msgs::PointCloud2::ConstPtr& msg);     // iterative_closest_point_node.cpp

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
    msg->data.assign((size_t)n * 32, 0);
    for (int i = 0; i < n; ++i) {
        std::memcpy(&msg->data[(size_t)i * 32], &raw[4 * (size_t)i], 12);
        std::memcpy(&msg->data[(size_t)i * 32 + 16], &raw[4 * (size_t)i + 3], 4);
    }
    // launch values: ground_plane_segmentation.launch:14-22, iterative_closest_point.launch:31-43
    ros::stub::set("voxel_size", 0.005);
    ros::stub::set("distance_threshold", 0.015);
    ros::stub::set("invert", true);
    ros::stub::set("template_cuboid_path", argv[2]);
    ros::stub::set("length", 0.2);
    ros::stub::set("width", 0.1);
    ros::stub::set("height", 0.03);
    ros::stub::set("icp_fitness_score", argc > 3 ? std::atof(argv[3]) : 0.0004);
    if (gps_node_main(argc, argv) != 0) return 3;
    if (icp_node_main(argc, argv) != 0) return 4;
    callback(msg);
    const auto* pts = ros::stub::last<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points");
    if (!pts) { std::fprintf(stderr, "gps published nothing\n"); return 5; }
    icp_callback(std::make_shared<sensor_msgs::PointCloud2>(*pts));
    std::printf("gps_points %u point_step %u\n", pts->width * pts->height, pts->point_step);
    const auto* pose = ros::stub::last<geometry_msgs::Pose>("/icp/pose");
    const auto* bbox = ros::stub::last<sensor_msgs::PointCloud2>("/icp/bbox_points");
    const auto* al = ros::stub::last<sensor_msgs::PointCloud2>("/icp/aligned_points");
    const auto* tp = ros::stub::last<sensor_msgs::PointCloud2>("/icp/template");
    std::printf("published pose %d bbox %d aligned %d template %d tf %d\n", pose ? 1 : 0, bbox ? 1 : 0, al ? 1 : 0, tp ? 1 : 0, (int)tf::TransformBroadcaster::sent().size());
    if (pose) std::printf("pose %a %a %a quat %a %a %a %a\n", pose->position.x, pose->position.y, pose->position.z, pose->orientation.x,
                          pose->orientation.y, pose->orientation.z, pose->orientation.w);
    if (bbox) {
        std::printf("bbox");
        for (uint32_t i = 0; i < bbox->width * 3; ++i) { float v; std::memcpy(&v, &bbox->data[4 * (size_t)i], 4); std::printf(" %a", (double)v); }
        std::printf("\n");
    }
    if (al) std::printf("aligned_points %u\n", al->width);
    if (tp) std::printf("template_points %u\n", tp->width);
    if (!tf::TransformBroadcaster::sent().empty())
        std::printf("tf %s -> %s\n", tf::TransformBroadcaster::sent().back().frame_id.c_str(), tf::TransformBroadcaster::sent().back().child_frame_id.c_str());
    // a second frame: the node has latched its result (icp.cpp:139-147) and only republishes
    const int before = ros::stub::count("/icp/pose");
    icp_callback(std::make_shared<sensor_msgs::PointCloud2>(*pts));
    std::printf("republished %d aligned_again %d\n", ros::stub::count("/icp/pose") - before, ros::stub::count("/icp/aligned_points"));
    return 0;
}
