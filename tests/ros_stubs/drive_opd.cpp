// Drives the object_pose_detection node shim without ROS: a frame arrives on the input topic (pcl_callback), the
// detect_objects service is called for object ids given on the command line, another frame arrives (after a success the
// node re-publishes the registered cloud, the template, the pose, the grasp marker and the TF on EVERY frame,
// opd.cpp:257-267).  Prints what the node registered and published; the clouds go to <out_dir>/<id>_<topic>.bin as
// int32 {point_step, width, height, n_fields, is_dense, row_step}, per field {offset, datatype, count, name[16]}, frame_id[64], data.
//   opd_shim_driver frame.bin template_dir/ voxel_size distance_threshold out_dir id [id ...]
#define main opd_node_main
#include "ros/object_pose_detection_node.cpp"
#undef main
#include <cstdio>

static void dump_cloud(const std::string& path, const sensor_msgs::PointCloud2& m) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return;
    const int32_t head[6] = {(int32_t)m.point_step, (int32_t)m.width, (int32_t)m.height, (int32_t)m.fields.size(), m.is_dense ? 1 : 0, (int32_t)m.row_step};
    std::fwrite(head, 4, 6, f);
    for (const auto& pf : m.fields) {
        const int32_t d[3] = {(int32_t)pf.offset, (int32_t)pf.datatype, (int32_t)pf.count};
        char nm[16] = {0};
        std::snprintf(nm, sizeof(nm), "%s", pf.name.c_str());
        std::fwrite(d, 4, 3, f);
        std::fwrite(nm, 1, 16, f);
    }
    char frame[64] = {0};
    std::snprintf(frame, sizeof(frame), "%s", m.header.frame_id.c_str());
    std::fwrite(frame, 1, 64, f);
    std::fwrite(m.data.data(), 1, m.data.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 7) return 2;
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
    msg->header.seq = 17;
    msg->height = 480; msg->width = (uint32_t)(n / 480);
    msg->point_step = 32; msg->row_step = 32 * msg->width; msg->is_dense = false;
    const char* names[4] = {"x", "y", "z", "rgb"};
    const uint32_t offs[4] = {0, 4, 8, 16};
    for (int k = 0; k < 4; ++k) { sensor_msgs::PointField pf; pf.name = names[k]; pf.offset = offs[k]; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; msg->fields.push_back(pf); }
    msg->data.assign((size_t)n * 32, 0xAB);   // junk in the input's padding must not reach the output
    for (int i = 0; i < n; ++i) {
        std::memcpy(&msg->data[(size_t)i * 32], &raw[4 * (size_t)i], 12);
        std::memcpy(&msg->data[(size_t)i * 32 + 16], &raw[4 * (size_t)i + 3], 4);
    }
    ros::stub::set("invert", true);
    ros::stub::set("voxel_size", std::atof(argv[3]));
    ros::stub::set("distance_threshold", std::atof(argv[4]));
    ros::stub::set("icp_fitness_score", 0.0004);
    ros::stub::set("template_path", argv[2]);
    const std::string out_dir = argv[5];
    if (opd_node_main(argc, argv) != 0) return 3;
    // what main() registered (opd.cpp:476-485)
    std::printf("registered sub_input %d service %d output %d registered_pcl %d bbox %d template %d pose %d marker %d n_adv %d n_sub %d\n",
                ros::stub::subscribes<sensor_msgs::PointCloud2>("/camera/depth/color/points", 1) ? 1 : 0,
                (ros::stub::services().size() == 1 && ros::stub::services()[0] == "detect_objects") ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/object_pose_detection/points") ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/registered_pcl") ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/bbox_points") ? 1 : 0,
                ros::stub::advertises<sensor_msgs::PointCloud2>("/icp/template") ? 1 : 0,
                ros::stub::advertises<geometry_msgs::Pose>("/icp/pose") ? 1 : 0,
                ros::stub::advertises<visualization_msgs::Marker>("object_pose_detection/grasp_pose") ? 1 : 0,
                (int)ros::stub::advertised().size(), (int)ros::stub::subscribed().size());
    const char* topics[6] = {"/object_pose_detection/points", "/icp/registered_pcl", "/icp/template", "/icp/pose", "object_pose_detection/grasp_pose", "/icp/bbox_points"};
    for (int a = 6; a < argc; ++a) {
        pcl_callback(msg);
        object_detection::ObjectDetection::Request req;
        object_detection::ObjectDetection::Response res;
        req.object_id = (uint8_t)std::atoi(argv[a]);
        int before[6];
        for (int k = 0; k < 6; ++k) before[k] = ros::stub::count(topics[k]);
        const size_t tf_before = tf::TransformBroadcaster::sent().size();
        const bool ret = service_callback(req, res);
        std::printf("service id %d returned %d success %d output_published %d\n", (int)req.object_id, ret ? 1 : 0, res.success ? 1 : 0,
                    ros::stub::count(topics[0]) - before[0]);
        const std::string tag = out_dir + "/" + std::to_string((int)req.object_id) + "_";
        if (const auto* out = ros::stub::last<sensor_msgs::PointCloud2>(topics[0])) dump_cloud(tag + "output.bin", *out);
        if (res.success) {
            std::printf("chosen size %d iterations %d accepted %d fitness %a\n", chosen.size, chosen.iterations, chosen.accepted, chosen.fitness);
            std::printf("T");
            for (int i = 0; i < 16; ++i) std::printf(" %a", (double)chosen.T[i]);
            std::printf("\n");
        }
        for (int rep = 0; rep < 2; ++rep) pcl_callback(msg);   // two more frames: everything cached goes out once per frame
        std::printf("per_two_frames registered_pcl %d template %d pose %d marker %d bbox %d tf %d\n", ros::stub::count(topics[1]) - before[1],
                    ros::stub::count(topics[2]) - before[2], ros::stub::count(topics[3]) - before[3], ros::stub::count(topics[4]) - before[4],
                    ros::stub::count(topics[5]) - before[5], (int)(tf::TransformBroadcaster::sent().size() - tf_before));
        if (ros::stub::count(topics[3]) > before[3]) {
            const auto* pose = ros::stub::last<geometry_msgs::Pose>(topics[3]);
            std::printf("pose %a %a %a quat %a %a %a %a\n", pose->position.x, pose->position.y, pose->position.z, pose->orientation.x,
                        pose->orientation.y, pose->orientation.z, pose->orientation.w);
            const auto* mk = ros::stub::last<visualization_msgs::Marker>(topics[4]);
            std::printf("marker frame %s ns %s id %d type %d action %d pose_equal %d scale %a %a %a color %a %a %a %a\n", mk->header.frame_id.c_str(), mk->ns.c_str(), mk->id,
                        mk->type, mk->action,
                        (mk->pose.position.x == pose->position.x && mk->pose.position.y == pose->position.y && mk->pose.position.z == pose->position.z &&
                         mk->pose.orientation.x == pose->orientation.x && mk->pose.orientation.y == pose->orientation.y &&
                         mk->pose.orientation.z == pose->orientation.z && mk->pose.orientation.w == pose->orientation.w) ? 1 : 0,
                        mk->scale.x, mk->scale.y, mk->scale.z, (double)mk->color.r, (double)mk->color.g, (double)mk->color.b, (double)mk->color.a);
            const auto& t = tf::TransformBroadcaster::sent().back();
            std::printf("tf %s -> %s origin_equal %d\n", t.frame_id.c_str(), t.child_frame_id.c_str(),
                        (t.t.x == pose->position.x && t.t.y == pose->position.y && t.t.z == pose->position.z && t.q.x == pose->orientation.x && t.q.w == pose->orientation.w) ? 1 : 0);
            dump_cloud(tag + "registered.bin", *ros::stub::last<sensor_msgs::PointCloud2>(topics[1]));
            dump_cloud(tag + "template.bin", *ros::stub::last<sensor_msgs::PointCloud2>(topics[2]));
        }
    }
    return 0;
}
