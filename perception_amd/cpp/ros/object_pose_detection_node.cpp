// object_pose_detection node on the HIP path: same node name, private parameters (invert, voxel_size,
// distance_threshold, input, output, icp_fitness_score, template_path: opd.cpp:451-464), subscription
// (input topic, queue 1, opd.cpp:476), service ("detect_objects", object_detection/ObjectDetection,
// opd.cpp:477) and publications as object_detection/src/object_pose_detection.cpp:
//   <output>                              sensor_msgs/PointCloud2   opd.cpp:480, published per service call :341-343
//   /icp/registered_pcl                   sensor_msgs/PointCloud2   opd.cpp:481, per frame after a success :259-262
//   /icp/bbox_points                      sensor_msgs/PointCloud2   opd.cpp:482 (advertised; its publication is commented out, :265)
//   /icp/template                         sensor_msgs/PointCloud2   opd.cpp:483, per frame after a success :263-264
//   /icp/pose                             geometry_msgs/Pose        opd.cpp:484, :173
//   object_pose_detection/grasp_pose      visualization_msgs/Marker opd.cpp:485, :96-136
//   TF camera_depth_optical_frame -> object_frame                    opd.cpp:172
//
// service_callback (opd.cpp:270-442) becomes ONE compute call: cd_process_batch runs crop, voxel grid,
// plane, extract, second z crop, clustering and the ICP of every cluster against the requested
// template on the GPU; the clouds the node publishes are read back from the device afterwards
// (cd_get_frame_cloud, cd_get_cluster_points).  What stays here is the reference's bookkeeping around
// it: the template table indexed by object_id (opd.cpp:87-88), the pick of the cluster whose size is
// closest to the template's (NOT the best ICP fitness, opd.cpp:411-423, argmin starts at "no cluster"
// with score 1000), the success threshold of 250 points (opd.cpp:429) and the re-publication of the
// cached results on every frame after a success (opd.cpp:257-267).  Templates are uploaded once per
// object id instead of re-read per cluster (opd.cpp:396).  Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <geometry_msgs/Pose.h>
#include <object_detection/ObjectDetection.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_broadcaster.h>
#include <visualization_msgs/Marker.h>

#include "../pcl_compat.hpp"
#include "ros_msgs.hpp"

static const char* template_filenames[] = {"", "screwdriver_ascii_tf.pcd", "eraser_ascii_tf.pcd", "clamp_ascii_tf.pcd", "marker_ascii_tf.pcd"};
static ros::Publisher pcl_pub, icp_pub, bbox_pub, template_pub, pose_pub, marker_pub;
static bool invert = true, ICP_SUCCESS = false;
static double voxel_size = 0.01, distance_threshold = 0.01, icp_fitness_score = 0.0004;
static double dimensions[] = {0.02, 0.02, 0.15};           // opd.cpp:80
static std::string template_path;
static sensor_msgs::PointCloud2ConstPtr input_pcl;      // latest frame (opd.cpp:249-252)
static sensor_msgs::PointCloud2 output_msg, template_msg;  // cached for the per-frame re-publication (opd.cpp:75-76)
static int argmin = -1;                                  // opd.cpp:71
cd_cluster_result chosen;                                // transform of the picked cluster (icp_transform, opd.cpp:72,426)
static pclhip::PointCloud<pclhip::PointXYZ> loaded_template[CD_MAX_TEMPLATES];

static int field_offset(const sensor_msgs::PointCloud2& m, const char* name) {
    for (const auto& f : m.fields) if (f.name == name) return (int)f.offset;
    return -1;
}

static void publish_grasp_marker(const geometry_msgs::Pose& p) {   // opd.cpp:96-136
    visualization_msgs::Marker marker;
    marker.header.frame_id = "camera_depth_optical_frame";
    marker.header.stamp = ros::Time::now();
    marker.ns = "grasp_pose";
    marker.id = 0;
    marker.type = visualization_msgs::Marker::CUBE;
    marker.action = visualization_msgs::Marker::ADD;
    marker.pose = p;
    marker.scale.x = dimensions[0]; marker.scale.y = dimensions[1]; marker.scale.z = dimensions[2];
    marker.color.r = 1.0f; marker.color.g = 0.0f; marker.color.b = 0.0f; marker.color.a = 0.5f;
    marker.lifetime = ros::Duration();
    marker_pub.publish(marker);
}

static void publish_pose(const double H[16]) {            // opd.cpp:138-174
    double pos[3], q[4];
    cd_pose_to_position_quaternion(H, pos, q);             // tf::Matrix3x3::getRotation
    geometry_msgs::Pose p;
    p.position.x = pos[0]; p.position.y = pos[1]; p.position.z = pos[2];
    p.orientation.x = q[0]; p.orientation.y = q[1]; p.orientation.z = q[2]; p.orientation.w = q[3];
    publish_grasp_marker(p);
    static tf::TransformBroadcaster br;
    tf::Transform transform(tf::Quaternion(q[0], q[1], q[2], q[3]), tf::Vector3(pos[0], pos[1], pos[2]));
    br.sendTransform(tf::StampedTransform(transform, ros::Time::now(), "camera_depth_optical_frame", "object_frame"));
    pose_pub.publish(p);
}

void pcl_callback(const sensor_msgs::PointCloud2ConstPtr& input) {
    input_pcl = input;
    if (ICP_SUCCESS && argmin != -1) {                     // opd.cpp:257-267
        icp_pub.publish(output_msg);
        template_msg.header.frame_id = "camera_depth_optical_frame";
        template_pub.publish(template_msg);
        publish_pose(chosen.pose);
    }
}

bool service_callback(object_detection::ObjectDetection::Request& req, object_detection::ObjectDetection::Response& res) {
    res.success = false;
    const int id = (int)req.object_id;
    if (!input_pcl || id < 1 || id > 4 || id >= CD_MAX_TEMPLATES) return false;
    bool xyz_ok = input_pcl->point_step >= 12 && (input_pcl->point_step & 3u) == 0;
    for (const auto& f : input_pcl->fields)
        if ((f.name == "x" || f.name == "y" || f.name == "z") && f.datatype != sensor_msgs::PointField::FLOAT32) xyz_ok = false;
    if (!xyz_ok || field_offset(*input_pcl, "x") != 0 || field_offset(*input_pcl, "y") != 4 || field_offset(*input_pcl, "z") != 8) {
        ROS_ERROR("object_pose_detection: expected FLOAT32 x,y,z at byte offsets 0,4,8 of a record and a point_step that is a multiple of 4");
        return false;
    }
    const int n = (int)(input_pcl->width * input_pcl->height);
    cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
    bool have_template = loaded_template[id].size() > 0;
    if (!have_template) {
        pclhip::PointCloud<pclhip::PointXYZ> tpl;
        if (pclhip::io::loadPCDFile(template_path + template_filenames[id], tpl) != -1 && tpl.size() > 0 &&
            cd_set_template(ctx, id, tpl.points.data(), sizeof(pclhip::PointXYZ), (int)tpl.size()) == CD_OK) {
            loaded_template[id] = tpl;
            have_template = true;
        }
    }
    cd_params prm;
    cd_default_params(&prm);                               // crop limits, cluster 0.02/200/25000, ICP 5000/1e-9 as in opd.cpp
    prm.leaf_size = (float)voxel_size;
    prm.plane_distance_threshold = distance_threshold;
    prm.extract_negative = invert ? 1 : 0;
    prm.crop2_enable = 1;                                  // opd.cpp:331-336
    prm.cluster_enable = 1;
    prm.template_slot = id;                                // (an empty slot: the chain still runs, the ICPs report "no template")
    prm.rgb_offset = field_offset(*input_pcl, "rgb");
    prm.icp_euclidean_fitness_epsilon = icp_fitness_score;
    prm.icp_accept_fitness = icp_fitness_score;
    cd_frame_result r;
    if (cd_process_batch(ctx, input_pcl->data.data(), input_pcl->point_step, n, 1, &prm, &r, nullptr, nullptr) != CD_OK) { ROS_ERROR("%s", cd_last_error(ctx)); return false; }
    {
        // opd.cpp:340-343: the cloud after the plane removal and the second crop, in the INPUT's layout (PassThrough, VoxelGrid
        // and ExtractIndices on PCLPointCloud2 keep the field table and point_step; fromPCL copies the header)
        sensor_msgs::PointCloud2 output;
        output.header = input_pcl->header;
        output.height = 1;
        output.is_dense = true;
        output.is_bigendian = input_pcl->is_bigendian;
        output.fields = input_pcl->fields;
        output.point_step = input_pcl->point_step;
        output.data.assign((size_t)std::max(r.n_objects, 0) * output.point_step, 0);
        int got = 0;
        if (cd_get_frame_cloud(ctx, 0, CD_CLOUD_OBJECTS, output.data.data(), output.point_step, prm.rgb_offset >= 12 ? prm.rgb_offset : -1, r.n_objects, &got) != CD_OK) {
            ROS_ERROR("%s", cd_last_error(ctx));
            return false;
        }
        output.width = (uint32_t)got;
        output.row_step = output.width * output.point_step;
        pcl_pub.publish(output);
    }
    ICP_SUCCESS = false;                                   // opd.cpp:372-374
    argmin = -1;
    if (r.n_clusters > 0 && !have_template) {              // opd.cpp:398-402: the first cluster's loadPCDFile fails
        ROS_ERROR("Couldn't read the template PCL file");
        return false;
    }
    // opd.cpp:376-423: EVERY cluster was registered and the pick is over all of them; the record holds the
    // CD_MAX_CLUSTERS_PER_FRAME largest, a frame with more (CD_FRAME_MORE_CLUSTERS) hands out the rest on request
    std::vector<cd_cluster_result> all((size_t)std::max(r.n_clusters, 1));
    const int got = cd_get_cluster_results(ctx, 0, 0, r.n_clusters, all.data(), nullptr);
    if (got != r.n_clusters) { ROS_ERROR("cd_get_cluster_results: %d of %d clusters", got, r.n_clusters); return false; }
    long min_score = 1000;                                 // opd.cpp:416-423
    const long tpl_size = (long)loaded_template[id].size();
    for (int k = 0; k < got; ++k) {
        const long diff = std::labs((long)all[(size_t)k].size - tpl_size);
        if (diff < min_score) { argmin = k; min_score = diff; }
    }
    if (argmin < 0) return false;                          // the reference indexes icp_transforms[-1] here (undefined)
    chosen = all[(size_t)argmin];                          // opd.cpp:426 (one transform per cluster, not per retry)
    // output_pcls[argmin] (opd.cpp:243,259): the cloud icp.align returned for the picked cluster, as
    // pcl::toROSMsg(PointCloud<PointXYZ>) lays it out - 16-byte records - with the header align() copies from its input
    pclhip::toROSMsgXYZ(nullptr, chosen.size, input_pcl->header, output_msg);
    int na = 0;
    if (cd_get_cluster_points(ctx, 0, argmin, 1, output_msg.data.data(), 16, chosen.size, &na) != CD_OK || na != chosen.size) { ROS_ERROR("%s", cd_last_error(ctx)); return false; }
    // template_msg (opd.cpp:242): toROSMsg of the template as loaded (a PCD file carries no frame; set when published)
    pclhip::toROSMsgXYZ(&loaded_template[id], (int)loaded_template[id].size(), std_msgs::Header(), template_msg);
    ICP_SUCCESS = min_score < 250;                         // opd.cpp:429
    res.success = ICP_SUCCESS;
    return ICP_SUCCESS;
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "object_pose_detection");
    ros::NodeHandle nh("~");
    std::string input_topic = "/camera/depth/color/points", output_topic = "/object_pose_detection/points";
    nh.getParam("invert", invert);
    nh.getParam("voxel_size", voxel_size);
    nh.getParam("distance_threshold", distance_threshold);
    nh.getParam("input", input_topic);
    nh.getParam("output", output_topic);
    nh.getParam("icp_fitness_score", icp_fitness_score);
    nh.getParam("template_path", template_path);
    ros::Subscriber sub = nh.subscribe(input_topic, 1, pcl_callback);
    ros::ServiceServer service = nh.advertiseService("detect_objects", service_callback);
    pcl_pub = nh.advertise<sensor_msgs::PointCloud2>(output_topic, 1);
    icp_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/registered_pcl", 1);
    bbox_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/bbox_points", 1);
    template_pub = nh.advertise<sensor_msgs::PointCloud2>("/icp/template", 1);
    pose_pub = nh.advertise<geometry_msgs::Pose>("/icp/pose", 1);
    marker_pub = nh.advertise<visualization_msgs::Marker>("object_pose_detection/grasp_pose", 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }
#endif
