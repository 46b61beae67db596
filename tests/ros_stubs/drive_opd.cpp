// Drives the object_pose_detection node shim without ROS: a frame arrives on the input topic (pcl_callback), the
// detect_objects service is called for object ids given on the command line, another frame arrives (the node republishes
// the cached pose after a success, opd.cpp:257-267).
//   opd_shim_driver frame.bin template_dir/ voxel_size distance_threshold id [id ...]
#define main opd_node_main
#include "ros/object_pose_detection_node.cpp"
#undef main
#include <cstdio>

int main(int argc, char** argv) {
    if (argc < 6) return 2;
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
    ros::stub::set("invert", true);
    ros::stub::set("voxel_size", std::atof(argv[3]));
    ros::stub::set("distance_threshold", std::atof(argv[4]));
    ros::stub::set("icp_fitness_score", 0.0004);
    ros::stub::set("template_path", argv[2]);
    if (opd_node_main(argc, argv) != 0) return 3;
    for (int a = 5; a < argc; ++a) {
        pcl_callback(msg);
        object_detection::ObjectDetection::Request req;
        object_detection::ObjectDetection::Response res;
        req.object_id = (uint8_t)std::atoi(argv[a]);
        const int before = ros::stub::count("/icp/pose");
        const bool ret = service_callback(req, res);
        std::printf("service id %d returned %d success %d\n", (int)req.object_id, ret ? 1 : 0, res.success ? 1 : 0);
        if (res.success) {
            std::printf("chosen size %d iterations %d accepted %d fitness %a\n", chosen.size, chosen.iterations, chosen.accepted, chosen.fitness);
            std::printf("T");
            for (int i = 0; i < 16; ++i) std::printf(" %a", (double)chosen.T[i]);
            std::printf("\n");
        }
        pcl_callback(msg);   // next frame
        const auto* pose = ros::stub::last<geometry_msgs::Pose>("/icp/pose");
        std::printf("poses_published %d\n", ros::stub::count("/icp/pose") - before);
        if (pose && ros::stub::count("/icp/pose") > before)
            std::printf("pose %a %a %a quat %a %a %a %a\n", pose->position.x, pose->position.y, pose->position.z, pose->orientation.x,
                        pose->orientation.y, pose->orientation.z, pose->orientation.w);
    }
    return 0;
}
