// object_pose_detection node on the HIP path: same node name, private parameters (invert, voxel_size,
// distance_threshold, input, output, icp_fitness_score, template_path: opd.cpp:451-464), subscription
// (input topic, queue 1, opd.cpp:476), service ("detect_objects", object_detection/ObjectDetection,
// opd.cpp:477) and publications (/icp/pose, /icp/bbox_points, /icp/template, output topic:
// opd.cpp:480-485) as object_detection/src/object_pose_detection.cpp.
//
// service_callback (opd.cpp:270-442) becomes ONE library call: cd_process_batch runs crop, voxel grid,
// plane, extract, second z crop, clustering and the ICP of every cluster against the requested
// template on the GPU.  What stays here is the reference's bookkeeping around it: the template table
// indexed by object_id (opd.cpp:87-88), the pick of the cluster whose size is closest to the
// template's (NOT the best ICP fitness, opd.cpp:411-423, argmin starts at "no cluster" with score
// 1000), the success threshold of 250 points (opd.cpp:429) and the re-publication of the cached
// pose on every frame after a success (opd.cpp:257-267).  Templates are uploaded once per object id
// instead of re-read per cluster (opd.cpp:396).  Builds only where ROS exists.
#ifdef CUBOID_HIP_WITH_ROS
#include <geometry_msgs/Pose.h>
#include <object_detection/ObjectDetection.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "../pcl_compat.hpp"

static const char* template_filenames[] = {"", "screwdriver_ascii_tf.pcd", "eraser_ascii_tf.pcd", "clamp_ascii_tf.pcd", "marker_ascii_tf.pcd"};
static ros::Publisher pose_pub, template_pub;
static bool invert = true, ICP_SUCCESS = false;
static double voxel_size = 0.01, distance_threshold = 0.01, icp_fitness_score = 0.0004;
static std::string template_path;
static sensor_msgs::PointCloud2ConstPtr input_pcl;      // latest frame (opd.cpp:249-252)
static cd_cluster_result chosen;
static int loaded_template_size[CD_MAX_TEMPLATES] = {0};

static void publish_pose(const double H[16]) {            // opd.cpp:131-160
    double pos[3], q[4];
    cd_pose_to_position_quaternion(H, pos, q);
    geometry_msgs::Pose p;
    p.position.x = pos[0]; p.position.y = pos[1]; p.position.z = pos[2];
    p.orientation.x = q[0]; p.orientation.y = q[1]; p.orientation.z = q[2]; p.orientation.w = q[3];
    pose_pub.publish(p);
}

void pcl_callback(const sensor_msgs::PointCloud2ConstPtr& input) {
    input_pcl = input;
    if (ICP_SUCCESS) publish_pose(chosen.pose);
}

bool service_callback(object_detection::ObjectDetection::Request& req, object_detection::ObjectDetection::Response& res) {
    res.success = false;
    const int id = (int)req.object_id;
    if (!input_pcl || id < 1 || id > 4 || id >= CD_MAX_TEMPLATES) return false;
    const int n = (int)(input_pcl->width * input_pcl->height);
    cd_context* ctx = pclhip::Device::instance(std::max(n, 640 * 480)).ctx();
    if (loaded_template_size[id] == 0) {
        pclhip::PointCloud<pclhip::PointXYZ> tpl;
        if (pclhip::io::loadPCDFile(template_path + template_filenames[id], tpl) == -1) { ROS_ERROR("Couldn't read the template PCL file"); return false; }
        if (cd_set_template(ctx, id, tpl.points.data(), sizeof(pclhip::PointXYZ), (int)tpl.size()) != CD_OK) { ROS_ERROR("%s", cd_last_error(ctx)); return false; }
        loaded_template_size[id] = (int)tpl.size();
    }
    cd_params prm;
    cd_default_params(&prm);                               // crop limits, cluster 0.02/200/25000, ICP 5000/1e-9 as in opd.cpp
    prm.leaf_size = (float)voxel_size;
    prm.plane_distance_threshold = distance_threshold;
    prm.extract_negative = invert ? 1 : 0;
    prm.crop2_enable = 1;                                  // opd.cpp:331-336
    prm.cluster_enable = 1;
    prm.template_slot = id;
    prm.icp_euclidean_fitness_epsilon = icp_fitness_score;
    prm.icp_accept_fitness = icp_fitness_score;
    cd_frame_result r;
    if (cd_process_batch(ctx, input_pcl->data.data(), input_pcl->point_step, n, 1, &prm, &r, nullptr, nullptr) != CD_OK) { ROS_ERROR("%s", cd_last_error(ctx)); return false; }
    // opd.cpp:376-423: EVERY cluster was registered and the pick is over all of them; the record holds the
    // CD_MAX_CLUSTERS_PER_FRAME largest, a frame with more (CD_FRAME_MORE_CLUSTERS) hands out the rest on request
    std::vector<cd_cluster_result> all((size_t)std::max(r.n_clusters, 1));
    const int got = cd_get_cluster_results(ctx, 0, 0, r.n_clusters, all.data(), nullptr);
    if (got != r.n_clusters) { ROS_ERROR("cd_get_cluster_results: %d of %d clusters", got, r.n_clusters); return false; }
    long min_score = 1000;                                 // opd.cpp:416-423
    int argmin = -1;
    for (int k = 0; k < got; ++k) {
        const long diff = std::labs((long)all[(size_t)k].size - (long)loaded_template_size[id]);
        if (diff < min_score) { argmin = k; min_score = diff; }
    }
    if (argmin < 0) { ICP_SUCCESS = false; return false; } // the reference indexes icp_transforms[-1] here (undefined)
    chosen = all[(size_t)argmin];
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
    pose_pub = nh.advertise<geometry_msgs::Pose>("/icp/pose", 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }
#endif
