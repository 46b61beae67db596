// surface_normal_estimation node on the HIP path: same node name, private parameters (invert,
// voxel_size, distance_threshold, input, output, plane_coefficients: sne.cpp:243-256), subscriptions
// (input cloud + the table plane's pcl_msgs/ModelCoefficients, :267-268) and publications
// (/surface_segmentation/normal_{x,y,z}_coefficients, /surface_segmentation/pose, TF
// camera_depth_optical_frame -> estimated_cuboid_frame: :271-275, :95) as
// cuboid_detection/src/surface_normal_estimation.cpp.  The callback (:167-234) is ONE library call:
// cd_surface_frame runs the three axis-constrained RANSAC fits on the GPU and assembles the frame.
// Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <geometry_msgs/Pose.h>
#include <pcl_msgs/ModelCoefficients.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

#include "../pcl_compat.hpp"

static ros::Publisher normal_x_pub, normal_y_pub, normal_z_pub, pose_pub;
static bool invert = true, coefficients_set = false;
static double voxel_size = 0.01, distance_threshold = 0.01;
static float table_normal[3] = {0.f, 0.f, 1.f};

void coefficients_callback(const pcl_msgs::ModelCoefficients& input) {   // sne.cpp:99-103
    if (input.values.size() < 3) return;
    for (int i = 0; i < 3; ++i) table_normal[i] = input.values[i];
    coefficients_set = true;
}

static pcl_msgs::ModelCoefficients coeff_msg(const std_msgs::Header& h, const float c[4]) {
    pcl_msgs::ModelCoefficients m;
    m.header = h;
    m.values.assign(c, c + 4);
    return m;
}

void callback(const sensor_msgs::PointCloud2ConstPtr& input) {
    if (!coefficients_set) return;                                       // sne.cpp:170
    const int n = (int)(input->width * input->height);
    cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
    cd_params prm;
    cd_default_params(&prm);
    prm.plane_distance_threshold = distance_threshold;
    cd_surface_frame_result r;
    if (cd_surface_frame(ctx, input->data.data(), input->point_step, n, table_normal, invert ? 1 : 0, &prm, &r) != CD_OK) return;
    double H[16], pos[3], q[4];
    for (int i = 0; i < 16; ++i) H[i] = r.Rt[i];
    cd_pose_to_position_quaternion(H, pos, q);                           // convert_eigen_to_tf, sne.cpp:63-97
    geometry_msgs::Pose p;
    p.position.x = pos[0]; p.position.y = pos[1]; p.position.z = pos[2];
    p.orientation.x = q[0]; p.orientation.y = q[1]; p.orientation.z = q[2]; p.orientation.w = q[3];
    static tf::TransformBroadcaster br;
    tf::Transform t(tf::Quaternion(q[0], q[1], q[2], q[3]), tf::Vector3(pos[0], pos[1], pos[2]));
    br.sendTransform(tf::StampedTransform(t, ros::Time::now(), "camera_depth_optical_frame", "estimated_cuboid_frame"));
    pose_pub.publish(p);
    normal_x_pub.publish(coeff_msg(input->header, r.coeff[2]));          // sne.cpp:231-233
    normal_y_pub.publish(coeff_msg(input->header, r.coeff[1]));
    normal_z_pub.publish(coeff_msg(input->header, r.coeff[0]));
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "surface_normal_estimation");
    ros::NodeHandle nh("~");
    std::string input_topic = "/ground_plane_segmentation/points", coefficients_topic = "/ground_plane_segmentation/coefficients";
    nh.getParam("invert", invert);
    nh.getParam("voxel_size", voxel_size);
    nh.getParam("distance_threshold", distance_threshold);
    nh.getParam("input", input_topic);
    nh.getParam("plane_coefficients", coefficients_topic);
    ros::Subscriber pcl_sub = nh.subscribe(input_topic, 1, callback);
    ros::Subscriber coef_sub = nh.subscribe(coefficients_topic, 1, coefficients_callback);
    normal_x_pub = nh.advertise<pcl_msgs::ModelCoefficients>("/surface_segmentation/normal_x_coefficients", 1);
    normal_y_pub = nh.advertise<pcl_msgs::ModelCoefficients>("/surface_segmentation/normal_y_coefficients", 1);
    normal_z_pub = nh.advertise<pcl_msgs::ModelCoefficients>("/surface_segmentation/normal_z_coefficients", 1);
    pose_pub = nh.advertise<geometry_msgs::Pose>("/surface_segmentation/pose", 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }
#endif
