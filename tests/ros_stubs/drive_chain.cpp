// Drives the ground_plane_segmentation and iterative_closest_point node shims back to back without ROS, the way
// iterative_closest_point.launch wires them (gps -> /ground_plane_segmentation/points -> icp): a raw x,y,z,rgb float32
// frame goes into the gps callback as a 32-byte-record PointCloud2, what gps published goes into the icp callback, and
// what icp published (/icp/pose, /icp/bbox_points, /icp/aligned_points, /icp/template, TF) is dumped as text.
//   chain_shim_driver frame.bin template.pcd [icp_fitness_score [use_surface_pose px py pz qx qy qz qw [clouds.bin]]]
// The two nodes are separate translation units (gps_node.o / icp_node.o, main renamed); this file only sees their entry points.
#include <algorithm>
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
void pose_callback(const geometry_msgs::Pose::ConstPtr& msg);         // iterative_closest_point_node.cpp

// pcl::toROSMsg(PointCloud<PointXYZ>) on the wire: 16-byte records x y z 1.0f, three float32 fields at 0 / 4 / 8, height 1
static int xyz16_layout_ok(const sensor_msgs::PointCloud2* m) {
    if (!m || m->point_step != 16 || m->row_step != 16 * m->width || m->height != 1 || !m->is_dense || m->is_bigendian || m->fields.size() != 3 ||
        m->data.size() != (size_t)16 * m->width)
        return 0;
    const char* names[3] = {"x", "y", "z"};
    for (int k = 0; k < 3; ++k)
        if (m->fields[k].name != names[k] || m->fields[k].offset != 4u * k || m->fields[k].datatype != sensor_msgs::PointField::FLOAT32 || m->fields[k].count != 1) return 0;
    for (uint32_t i = 0; i < m->width; ++i) { float w; std::memcpy(&w, &m->data[(size_t)i * 16 + 12], 4); if (w != 1.0f) return 0; }
    return 1;
}

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
    const bool use_pose = argc > 4 && std::atoi(argv[4]) != 0;   // the opt-in of icp.cpp:165-167
    if (use_pose) ros::stub::set("use_surface_pose", true);
    if (gps_node_main(argc, argv) != 0) return 3;
    if (icp_node_main(argc, argv) != 0) return 4;
    // what the two mains registered (gps.cpp:146-150, icp.cpp:226-233)
    std::printf("registered gps_sub %d gps_points %d gps_coeff %d icp_sub_points %d icp_sub_pose %d aligned %d bbox %d template %d pose %d n_adv %d n_sub %d\n",
                ros::stub::subscribes<sensor_msgs::PointCloud2>("/camera/depth/color/points", 1) ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points") ? 1 : 0,
                (int)std::count_if(ros::stub::advertised().begin(), ros::stub::advertised().end(), [](const ros::stub::Endpoint& e) { return e.name == "/ground_plane_segmentation/coefficients"; }),
                ros::stub::subscribes<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points", 1) ? 1 : 0,
                ros::stub::subscribes<geometry_msgs::Pose>("/surface_segmentation/pose", 1) ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/aligned_points") ? 1 : 0, ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/bbox_points") ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/template") ? 1 : 0, ros::stub::advertises<geometry_msgs::Pose>("/icp/pose") ? 1 : 0,
                (int)ros::stub::advertised().size(), (int)ros::stub::subscribed().size());
    callback(msg);
    const auto* pts = ros::stub::last<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points");
    if (!pts) { std::fprintf(stderr, "gps published nothing\n"); return 5; }
    if (use_pose) {
        // without a pose the node waits (icp.cpp:166) ...
        icp_callback(std::make_shared<sensor_msgs::PointCloud2>(*pts));
        std::printf("before_pose published %d\n", ros::stub::count("/icp/pose"));
        // ... then registers against the template moved by it (icp.cpp:167): argv[5..11] = position, quaternion
        auto pm = std::make_shared<geometry_msgs::Pose>();
        pm->position.x = std::atof(argv[5]); pm->position.y = std::atof(argv[6]); pm->position.z = std::atof(argv[7]);
        pm->orientation.x = std::atof(argv[8]); pm->orientation.y = std::atof(argv[9]); pm->orientation.z = std::atof(argv[10]); pm->orientation.w = std::atof(argv[11]);
        pose_callback(pm);
    }
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
        for (size_t i = 0; i + 4 <= bbox->data.size(); i += 4) { float v; std::memcpy(&v, &bbox->data[i], 4); std::printf(" %a", (double)v); }   // whole records
        std::printf("\n");
    }
    if (al) std::printf("aligned_points %u\n", al->width);
    if (tp) std::printf("template_points %u\n", tp->width);
    std::printf("layout aligned %d template %d bbox %d frames %s|%s|%s\n", xyz16_layout_ok(al), xyz16_layout_ok(tp), xyz16_layout_ok(bbox),
                al ? al->header.frame_id.c_str() : "-", tp ? tp->header.frame_id.c_str() : "-", bbox ? bbox->header.frame_id.c_str() : "-");
    if (argc > 12 && al) {   // the aligned cloud and the template as published, for the test to compare with the oracle
        FILE* f = std::fopen(argv[12], "wb");
        if (f) { std::fwrite(al->data.data(), 1, al->data.size(), f); std::fwrite(tp->data.data(), 1, tp->data.size(), f); std::fclose(f); }
    }
    if (!tf::TransformBroadcaster::sent().empty())
        std::printf("tf %s -> %s\n", tf::TransformBroadcaster::sent().back().frame_id.c_str(), tf::TransformBroadcaster::sent().back().child_frame_id.c_str());
    // two more frames: the node has latched its result and re-publishes ALL FOUR messages and the TF per frame (icp.cpp:139-147)
    const char* topics[4] = {"/icp/pose", "/icp/aligned_points", "/icp/template", "/icp/bbox_points"};
    int before[4];
    for (int k = 0; k < 4; ++k) before[k] = ros::stub::count(topics[k]);
    const size_t tf_before = tf::TransformBroadcaster::sent().size();
    const std::vector<uint8_t> al_data = al ? al->data : std::vector<uint8_t>();
    for (int rep = 0; rep < 2; ++rep) icp_callback(std::make_shared<sensor_msgs::PointCloud2>(*pts));
    const auto* al2 = ros::stub::last<sensor_msgs::PointCloud2>("/icp/aligned_points");
    std::printf("republished pose %d aligned %d template %d bbox %d tf %d same_aligned %d\n", ros::stub::count(topics[0]) - before[0], ros::stub::count(topics[1]) - before[1],
                ros::stub::count(topics[2]) - before[2], ros::stub::count(topics[3]) - before[3], (int)(tf::TransformBroadcaster::sent().size() - tf_before),
                (al2 && al2->data == al_data) ? 1 : 0);
    return 0;
}
