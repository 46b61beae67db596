// Drives the bbox_filter and surface_normal_estimation node shims without ROS.
//   bbox_sne_shim_driver bbox cloud.bin out.bin P0..P11 x1 y1 x2 y2
//       cloud.bin: x,y,z,rgb float32 records (16 B).  The cloud arrives once before any CameraInfo (everything rejected,
//       bbox_filter.cpp:33-34), then CameraInfo + Rectangle arrive and the cloud again; out.bin = the second publication's blob.
//   bbox_sne_shim_driver sne cloud.bin nx ny nz distance_threshold
//       cloud.bin: x,y,z float32 records (12 B); the table normal arrives on the coefficients topic first.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cuboid_detection/Rectangle.h>
#include <geometry_msgs/Pose.h>
#include <pcl_msgs/ModelCoefficients.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

int bbox_node_main(int argc, char** argv);
int sne_node_main(int argc, char** argv);
void info_cb(const sensor_msgs::CameraInfoConstPtr& msg);
void bbox_cb(const cuboid_detection::Rectangle::ConstPtr& msg);
void pcl_cb(const sensor_msgs::PointCloud2ConstPtr& input);
void coefficients_callback(const pcl_msgs::ModelCoefficients& input);
void callback(const sensor_msgs::PointCloud2ConstPtr& input);   // surface_normal_estimation_node.cpp

static std::shared_ptr<sensor_msgs::PointCloud2> load(const char* path, int rec_floats, bool with_rgb) {
    std::vector<float> raw;
    FILE* f = std::fopen(path, "rb");
    if (!f) return nullptr;
    float v;
    while (std::fread(&v, 4, 1, f) == 1) raw.push_back(v);
    std::fclose(f);
    const int n = (int)(raw.size() / rec_floats);
    auto msg = std::make_shared<sensor_msgs::PointCloud2>();
    msg->header.frame_id = "camera_depth_optical_frame";
    msg->height = 1; msg->width = (uint32_t)n;
    msg->point_step = 4u * (uint32_t)rec_floats; msg->row_step = msg->point_step * msg->width; msg->is_dense = true;
    const char* names[4] = {"x", "y", "z", "rgb"};
    for (int k = 0; k < (with_rgb ? 4 : 3); ++k) { sensor_msgs::PointField pf; pf.name = names[k]; pf.offset = 4u * k; pf.datatype = sensor_msgs::PointField::FLOAT32; pf.count = 1; msg->fields.push_back(pf); }
    msg->data.resize(raw.size() * 4);
    std::memcpy(msg->data.data(), raw.data(), raw.size() * 4);
    return msg;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "bbox") {
        if (argc < 20) return 2;
        auto cloud = load(argv[2], 4, true);
        if (!cloud) return 2;
        if (bbox_node_main(argc, argv) != 0) return 3;
        pcl_cb(cloud);
        const auto* first = ros::stub::last<sensor_msgs::PointCloud2>("/bbox_filter/points");
        std::printf("before_camera_info width %u\n", first ? first->width : 0xffffffffu);
        auto info = std::make_shared<sensor_msgs::CameraInfo>();
        for (int i = 0; i < 12; ++i) info->P[i] = std::atof(argv[4 + i]);
        info_cb(info);
        auto rect = std::make_shared<cuboid_detection::Rectangle>();
        rect->x1 = std::atoi(argv[16]); rect->y1 = std::atoi(argv[17]); rect->x2 = std::atoi(argv[18]); rect->y2 = std::atoi(argv[19]);
        bbox_cb(rect);
        pcl_cb(cloud);
        const auto* out = ros::stub::last<sensor_msgs::PointCloud2>("/bbox_filter/points");
        if (!out) return 4;
        std::printf("after width %u point_step %u fields %zu publications %d\n", out->width, out->point_step, out->fields.size(), ros::stub::count("/bbox_filter/points"));
        FILE* f = std::fopen(argv[3], "wb");
        if (!f) return 2;
        std::fwrite(out->data.data(), 1, out->data.size(), f);
        std::fclose(f);
        return 0;
    }
    if (mode == "sne") {
        if (argc < 7) return 2;
        auto cloud = load(argv[2], 3, false);
        if (!cloud) return 2;
        ros::stub::set("invert", true);
        ros::stub::set("distance_threshold", std::atof(argv[6]));
        if (sne_node_main(argc, argv) != 0) return 3;
        callback(cloud);   // no table plane yet: nothing may be published (sne.cpp:170)
        std::printf("before_coefficients poses %d\n", ros::stub::count("/surface_segmentation/pose"));
        pcl_msgs::ModelCoefficients co;
        co.values = {(float)std::atof(argv[3]), (float)std::atof(argv[4]), (float)std::atof(argv[5]), 0.f};
        coefficients_callback(co);
        callback(cloud);
        const auto* pose = ros::stub::last<geometry_msgs::Pose>("/surface_segmentation/pose");
        std::printf("after poses %d tf %zu\n", ros::stub::count("/surface_segmentation/pose"), tf::TransformBroadcaster::sent().size());
        if (pose) std::printf("pose %a %a %a quat %a %a %a %a\n", pose->position.x, pose->position.y, pose->position.z, pose->orientation.x,
                              pose->orientation.y, pose->orientation.z, pose->orientation.w);
        const char* topics[3] = {"/surface_segmentation/normal_x_coefficients", "/surface_segmentation/normal_y_coefficients", "/surface_segmentation/normal_z_coefficients"};
        for (int k = 0; k < 3; ++k) {
            const auto* c = ros::stub::last<pcl_msgs::ModelCoefficients>(topics[k]);
            std::printf("normal_%c", "xyz"[k]);
            if (c) for (float v : c->values) std::printf(" %a", (double)v);
            std::printf("\n");
        }
        if (!tf::TransformBroadcaster::sent().empty())
            std::printf("tf %s -> %s\n", tf::TransformBroadcaster::sent().back().frame_id.c_str(), tf::TransformBroadcaster::sent().back().child_frame_id.c_str());
        return 0;
    }
    return 2;
}
