// iterative_closest_point node on the HIP path: same node name, params (template_cuboid_path,
// length, width, height, icp_fitness_score: icp.cpp:212-216), subscription
// (/ground_plane_segmentation/points, icp.cpp:226) and publications (/icp/aligned_points,
// /icp/bbox_points, /icp/template, /icp/pose + TF camera_depth_optical_frame -> icp_cuboid_frame,
// icp.cpp:230-233,86) as cuboid_detection/src/iterative_closest_point.cpp.  The template is read
// once at start-up instead of once per frame (icp.cpp:159).  Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <geometry_msgs/Pose.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

#include "../pcl_compat.hpp"

static ros::Publisher pcl_pub, bbox_pub, template_pub, pose_pub;
static double dimensions[3], icp_fitness_score;
static bool ICP_SUCCESS = false;
static cd_cluster_result latched;
static pclhip::PointCloud<pclhip::PointXYZ> tpl;

static sensor_msgs::PointCloud2 xyz_msg(const float* xyz, int n, const std_msgs::Header& h) {
    sensor_msgs::PointCloud2 m;
    m.header = h;
    m.header.frame_id = "camera_depth_optical_frame";
    m.height = 1; m.width = (uint32_t)n; m.point_step = 12; m.row_step = 12u * (uint32_t)n; m.is_dense = true;
    m.fields.resize(3);
    const char* names[3] = {"x", "y", "z"};
    for (int k = 0; k < 3; ++k) { m.fields[k].name = names[k]; m.fields[k].offset = 4 * k; m.fields[k].datatype = sensor_msgs::PointField::FLOAT32; m.fields[k].count = 1; }
    m.data.resize((size_t)n * 12);
    std::memcpy(m.data.data(), xyz, (size_t)n * 12);
    return m;
}

static void publish_results(const std_msgs::Header& h) {
    double pos[3], q[4];
    float box[24];
    cd_pose_to_position_quaternion(latched.pose, pos, q);                       // icp.cpp:55-82
    cd_bbox_corners(latched.pose, dimensions[0], dimensions[1], dimensions[2], box);   // icp.cpp:90-110
    bbox_pub.publish(xyz_msg(box, 8, h));
    geometry_msgs::Pose p;
    p.position.x = pos[0]; p.position.y = pos[1]; p.position.z = pos[2];
    p.orientation.x = q[0]; p.orientation.y = q[1]; p.orientation.z = q[2]; p.orientation.w = q[3];
    static tf::TransformBroadcaster br;
    tf::Transform t(tf::Quaternion(q[0], q[1], q[2], q[3]), tf::Vector3(pos[0], pos[1], pos[2]));
    br.sendTransform(tf::StampedTransform(t, ros::Time::now(), "camera_depth_optical_frame", "icp_cuboid_frame"));
    pose_pub.publish(p);
}

void icp_callback(const sensor_msgs::PointCloud2::ConstPtr& msg) {
    if (ICP_SUCCESS) { publish_results(msg->header); return; }                  // icp.cpp:139-147 latch
    const int n = (int)(msg->width * msg->height);
    cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
    cd_params prm;
    cd_default_params(&prm);
    prm.icp_euclidean_fitness_epsilon = icp_fitness_score;
    prm.icp_accept_fitness = icp_fitness_score;
    std::vector<float> aligned((size_t)std::max(n, 1) * 3);
    cd_cluster_result r;
    const int st = cd_icp(ctx, 0, msg->data.data(), msg->point_step, n, &prm, &r, aligned.data());
    if (st != CD_OK || !r.accepted) return;                                     // icp.cpp:182
    latched = r;
    ICP_SUCCESS = true;
    pcl_pub.publish(xyz_msg(aligned.data(), n, msg->header));
    std::vector<float> t((size_t)tpl.size() * 3);
    for (size_t i = 0; i < tpl.size(); ++i) { t[3 * i] = tpl.points[i].x; t[3 * i + 1] = tpl.points[i].y; t[3 * i + 2] = tpl.points[i].z; }
    template_pub.publish(xyz_msg(t.data(), (int)tpl.size(), msg->header));
    publish_results(msg->header);
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "iterative_closest_point");
    ros::NodeHandle nh("~");
    std::string template_cuboid_filename;
    nh.getParam("template_cuboid_path", template_cuboid_filename);
    nh.getParam("length", dimensions[0]);
    nh.getParam("width", dimensions[1]);
    nh.getParam("height", dimensions[2]);
    nh.getParam("icp_fitness_score", icp_fitness_score);
    if (pclhip::io::loadPCDFile(template_cuboid_filename, tpl) == -1) { ROS_ERROR("Couldn't read the template PCL file"); return 1; }
    cd_set_template(pclhip::Device::instance().ctx(), 0, tpl.points.data(), sizeof(pclhip::PointXYZ), (int)tpl.size());
    ros::Subscriber pcl_sub = nh.subscribe<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points", 1, icp_callback);
    pcl_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/aligned_points", 1);
    bbox_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/bbox_points", 1);
    template_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/template", 1);
    pose_pub = nh.advertise<geometry_msgs::Pose>("/icp/pose", 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }
#endif
