// iterative_closest_point node on the HIP path: same node name, params (template_cuboid_path,
// length, width, height, icp_fitness_score: icp.cpp:212-216), subscriptions and publications as
// cuboid_detection/src/iterative_closest_point.cpp:
//   sub  /ground_plane_segmentation/points  sensor_msgs/PointCloud2  q 1   icp.cpp:226
//   sub  /surface_segmentation/pose         geometry_msgs/Pose       q 1   icp.cpp:227 (stored; inert unless use_surface_pose)
//   pub  /icp/aligned_points                sensor_msgs/PointCloud2        icp.cpp:230, :197, latch :141
//   pub  /icp/bbox_points                   sensor_msgs/PointCloud2        icp.cpp:231, :127
//   pub  /icp/template                      sensor_msgs/PointCloud2        icp.cpp:232, :199, latch :143
//   pub  /icp/pose                          geometry_msgs/Pose             icp.cpp:233, :87
//   TF   camera_depth_optical_frame -> icp_cuboid_frame                    icp.cpp:86
// The XYZ clouds go out as pcl::toROSMsg(PointCloud<PointXYZ>) lays them out (16-byte records, ros_msgs.hpp).  After the
// first accepted registration the node only re-publishes its cached messages, all four of them (icp.cpp:139-147).
// The template is read once at start-up instead of once per frame (icp.cpp:159).
//
// Opt-in, private parameter `use_surface_pose` (default false = the reference as it runs): what icp.cpp:165-167 has
// commented out - wait for a pose from surface_normal_estimation and move the template into that frame before
// registering against it.  (The reference's pose_callback converts the wrong variable, icp.cpp:133; here the message
// that arrived is the one stored.)  Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <geometry_msgs/Pose.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>

#include "../pcl_compat.hpp"
#include "ros_msgs.hpp"

static ros::Publisher pcl_pub, bbox_pub, template_pub, pose_pub;
static double dimensions[3], icp_fitness_score;
static bool ICP_SUCCESS = false, POSE_FLAG = false, use_surface_pose = false;
static cd_cluster_result latched;                                  // icp_transform (icp.cpp:31) and what it came from
static sensor_msgs::PointCloud2 output_msg, template_msg;          // icp.cpp:34-35
static geometry_msgs::Pose pose_msg;                               // icp.cpp:37
static pclhip::PointCloud<pclhip::PointXYZ> tpl;

static std_msgs::Header camera_frame() {                           // toROSMsg of a header-less cloud, then frame_id set by hand
    std_msgs::Header h;
    h.frame_id = "camera_depth_optical_frame";
    return h;
}

static void publish_bounding_box() {                               // icp.cpp:90-128
    float box[24];
    cd_bbox_corners(latched.pose, dimensions[0], dimensions[1], dimensions[2], box);
    sensor_msgs::PointCloud2 bbox_msg;
    pclhip::toROSMsgXYZ3(box, 8, camera_frame(), bbox_msg);
    bbox_pub.publish(bbox_msg);
}

static void publish_pose() {                                       // icp.cpp:55-88
    double pos[3], q[4];
    cd_pose_to_position_quaternion(latched.pose, pos, q);
    geometry_msgs::Pose p;
    p.position.x = pos[0]; p.position.y = pos[1]; p.position.z = pos[2];
    p.orientation.x = q[0]; p.orientation.y = q[1]; p.orientation.z = q[2]; p.orientation.w = q[3];
    static tf::TransformBroadcaster br;
    tf::Transform t(tf::Quaternion(q[0], q[1], q[2], q[3]), tf::Vector3(pos[0], pos[1], pos[2]));
    br.sendTransform(tf::StampedTransform(t, ros::Time::now(), "camera_depth_optical_frame", "icp_cuboid_frame"));
    pose_pub.publish(p);
}

static void publish_all() {                                        // icp.cpp:141-145 and :197-201
    pcl_pub.publish(output_msg);
    template_msg.header.frame_id = "camera_depth_optical_frame";
    template_pub.publish(template_msg);
    publish_bounding_box();
    publish_pose();
}

void pose_callback(const geometry_msgs::Pose::ConstPtr& msg) {     // icp.cpp:130-134
    POSE_FLAG = true;
    pose_msg = *msg;
}

// icp.cpp:165-167 (opt-in): the template moved by the surface pose, tf::poseMsgToEigen(...).cast<float>() applied as
// pcl::transformPointCloud does
static bool move_template_by_pose(pclhip::PointCloud<pclhip::PointXYZ>& moved) {
    const double x = pose_msg.orientation.x, y = pose_msg.orientation.y, z = pose_msg.orientation.z, w = pose_msg.orientation.w;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    float Rf[9];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
    const float t[3] = {(float)pose_msg.position.x, (float)pose_msg.position.y, (float)pose_msg.position.z};
    moved = tpl;
    for (auto& p : moved.points) {
        const float px = p.x, py = p.y, pz = p.z;
        p.x = ((Rf[0] * px + Rf[1] * py) + Rf[2] * pz) + t[0];
        p.y = ((Rf[3] * px + Rf[4] * py) + Rf[5] * pz) + t[1];
        p.z = ((Rf[6] * px + Rf[7] * py) + Rf[8] * pz) + t[2];
    }
    return true;
}

void icp_callback(const sensor_msgs::PointCloud2::ConstPtr& msg) {
    if (ICP_SUCCESS) { publish_all(); return; }                                 // icp.cpp:139-147: compute ICP only once
    const int n = (int)(msg->width * msg->height);
    cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
    const pclhip::PointCloud<pclhip::PointXYZ>* target = &tpl;
    pclhip::PointCloud<pclhip::PointXYZ> moved;
    if (use_surface_pose) {
        if (!POSE_FLAG) return;                                                 // icp.cpp:166
        move_template_by_pose(moved);                                           // icp.cpp:167
        if (cd_set_template(ctx, 0, moved.points.data(), sizeof(pclhip::PointXYZ), (int)moved.size()) != CD_OK) { ROS_ERROR("%s", cd_last_error(ctx)); return; }
        target = &moved;
    }
    cd_params prm;
    cd_default_params(&prm);
    prm.icp_euclidean_fitness_epsilon = icp_fitness_score;
    prm.icp_accept_fitness = icp_fitness_score;
    std::vector<float> aligned((size_t)std::max(n, 1) * 3);
    cd_cluster_result r;
    const int st = cd_icp(ctx, 0, msg->data.data(), msg->point_step, n, &prm, &r, aligned.data());
    if (st != CD_OK) return;
    latched = r;                                                                // icp.cpp:179: icp_transform is set either way
    if (!r.accepted) return;                                                    // icp.cpp:182
    ICP_SUCCESS = true;
    pclhip::toROSMsgXYZ3(aligned.data(), n, msg->header, output_msg);           // icp.cpp:193 (align() copies its input's header)
    pclhip::toROSMsgXYZ(target, (int)target->size(), std_msgs::Header(), template_msg);   // icp.cpp:194
    publish_all();
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
    nh.getParam("use_surface_pose", use_surface_pose);
    if (pclhip::io::loadPCDFile(template_cuboid_filename, tpl) == -1) { ROS_ERROR("Couldn't read the template PCL file"); return 1; }
    cd_set_template(pclhip::Device::instance().ctx(), 0, tpl.points.data(), sizeof(pclhip::PointXYZ), (int)tpl.size());
    ros::Subscriber pcl_sub = nh.subscribe<sensor_msgs::PointCloud2>("/ground_plane_segmentation/points", 1, icp_callback);
    ros::Subscriber pose_sub = nh.subscribe<geometry_msgs::Pose>("/surface_segmentation/pose", 1, pose_callback);
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
