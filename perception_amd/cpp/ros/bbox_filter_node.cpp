// bbox_filter node on the HIP path: same node name, subscriptions (/object_detection/bbox
// cuboid_detection/Rectangle, /camera/color/camera_info, /ground_plane_segmentation/points:
// bbox_filter.cpp:118-120) and publication (/bbox_filter/points, :123) as
// cuboid_detection/src/bbox_filter.cpp.  The per-point projection test (within_bbox, :30-51) runs on
// the GPU through cd_bbox_filter; the kept points are copied out of the incoming PointCloud2 blob in
// index order, which is what the reference's ExtractIndices<PCLPointCloud2> does (:96-101).
// Until a CameraInfo message has arrived every point is rejected, as in the reference (:33-34).
// Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <cuboid_detection/Rectangle.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/PointCloud2.h>

#include "../pcl_compat.hpp"

static ros::Publisher pcl_pub;
static double proj_matrix[12];
static int32_t bbox[4] = {0, 0, 0, 0};
static bool camera_info_read = false;

void info_cb(const sensor_msgs::CameraInfoConstPtr& msg) {   // bbox_filter.cpp:56-65
    for (int i = 0; i < 12; ++i) proj_matrix[i] = msg->P[i];
    camera_info_read = true;
}

void bbox_cb(const cuboid_detection::Rectangle::ConstPtr& msg) {   // bbox_filter.cpp:67-76
    bbox[0] = (int32_t)msg->x1; bbox[1] = (int32_t)msg->y1; bbox[2] = (int32_t)msg->x2; bbox[3] = (int32_t)msg->y2;
}

void pcl_cb(const sensor_msgs::PointCloud2ConstPtr& input) {
    const int n = (int)(input->width * input->height);
    std::vector<int32_t> keep((size_t)std::max(n, 1));
    int kept = 0;
    if (camera_info_read && n > 0) {
        cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
        if (cd_bbox_filter(ctx, input->data.data(), input->point_step, n, proj_matrix, bbox, keep.data(), n, &kept) != CD_OK) {
            ROS_ERROR("%s", cd_last_error(ctx));
            return;
        }
    }
    sensor_msgs::PointCloud2 out;
    out.header = input->header;
    out.fields = input->fields;
    out.is_bigendian = input->is_bigendian;
    out.point_step = input->point_step;
    out.height = 1;
    out.width = (uint32_t)kept;
    out.row_step = out.point_step * out.width;
    out.is_dense = input->is_dense;
    out.data.resize((size_t)kept * out.point_step);
    for (int i = 0; i < kept; ++i)
        std::memcpy(&out.data[(size_t)i * out.point_step], &input->data[(size_t)keep[(size_t)i] * input->point_step], out.point_step);
    pcl_pub.publish(out);
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "bbox_filter");
    ros::NodeHandle nh;
    ros::Subscriber bbox_sub = nh.subscribe("/object_detection/bbox", 1, bbox_cb);
    ros::Subscriber info_sub = nh.subscribe("/camera/color/camera_info", 1, info_cb);
    ros::Subscriber pcl_sub = nh.subscribe("/ground_plane_segmentation/points", 1, pcl_cb);
    pcl_pub = nh.advertise<sensor_msgs::PointCloud2>("/bbox_filter/points", 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }
#endif
