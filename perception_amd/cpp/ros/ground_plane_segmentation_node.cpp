// ground_plane_segmentation node on the HIP path.  Same node name, private params, topics and
// message types as cuboid_detection/src/ground_plane_segmentation.cpp (params :122-135, subscribe
// :146, publishers :149-150), so it drops into cuboid_detection/launch/ground_plane_segmentation.launch
// unchanged.  Builds only where catkin/roscpp/pcl_conversions exist (not in this image):
//   add_executable(ground_plane_segmentation perception_amd/cpp/ros/ground_plane_segmentation_node.cpp)
//   target_link_libraries(ground_plane_segmentation ${catkin_LIBRARIES} cuboid_hip)
#ifdef CUBOID_HIP_WITH_ROS
#include <pcl_conversions/pcl_conversions.h>
#include <pcl_msgs/ModelCoefficients.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "../pcl_compat.hpp"

static ros::Publisher pcl_pub, coef_pub;
static bool invert;
static double voxel_size, distance_threshold;
static std::string input_topic, output_topic, coefficients_topic;

static int field_offset(const sensor_msgs::PointCloud2& m, const char* name) {
    for (const auto& f : m.fields) if (f.name == name) return (int)f.offset;
    return -1;
}
// x, y, z as FLOAT32 at byte offsets 0 / 4 / 8 and records of whole 4-byte words: what the device code reads and writes
static bool xyz_float32_at_0_4_8(const sensor_msgs::PointCloud2& m) {
    int seen = 0;
    for (const auto& f : m.fields) {
        const int want = f.name == "x" ? 0 : f.name == "y" ? 4 : f.name == "z" ? 8 : -1;
        if (want < 0) continue;
        if ((int)f.offset != want || f.datatype != sensor_msgs::PointField::FLOAT32) return false;
        ++seen;
    }
    return seen == 3 && m.point_step >= 12 && (m.point_step & 3u) == 0;
}

void callback(const sensor_msgs::PointCloud2ConstPtr& input) {
    // The PointCloud2 blob goes to the device as it is: x,y,z float32 at offsets 0/4/8 (what every PCL point type and the
    // D435 driver use), stride = point_step.
    if (!xyz_float32_at_0_4_8(*input)) {
        ROS_ERROR("ground_plane_segmentation: expected FLOAT32 x,y,z at byte offsets 0,4,8 of a record and a point_step that is a multiple of 4");
        return;
    }
    cd_context* ctx = pclhip::Device::instance((int)(input->width * input->height)).ctx();
    cd_params prm;
    cd_default_params(&prm);
    prm.leaf_size = (float)voxel_size;
    prm.plane_distance_threshold = distance_threshold;
    prm.extract_negative = invert ? 1 : 0;
    prm.crop2_enable = 0;
    prm.rgb_offset = field_offset(*input, "rgb");
    const int n = (int)(input->width * input->height);
    // ONE library call for the callback's body (gps.cpp:53-101): one upload of the blob, one download of the records to
    // publish.  ExtractIndices<PCLPointCloud2>(negative = invert) + fromPCL (gps.cpp:96-112): the published cloud carries the
    // INPUT's field table and point_step - VoxelGrid<PCLPointCloud2> and ExtractIndices<PCLPointCloud2> both copy them - with
    // one record per kept voxel: centroid at the x,y,z offsets, averaged colour at the rgb offset, padding bytes zero.
    // (PCL also averages any OTHER field as a float32; the D435 driver publishes none, and here they stay zero.)
    if (prm.rgb_offset < 0) prm.rgb_offset = field_offset(*input, "rgba");
    sensor_msgs::PointCloud2 out;
    out.header = input->header;
    out.height = 1;
    out.is_dense = true;
    out.is_bigendian = input->is_bigendian;
    out.fields = input->fields;
    out.point_step = input->point_step;
    out.data.assign((size_t)n * out.point_step, 0);
    float coeff[4] = {0, 0, 0, 0};
    int kept = 0, ni = 0;
    const int st = cd_ground_plane(ctx, input->data.data(), input->point_step, n, &prm, coeff, out.data.data(), n, &kept, &ni);
    if (st != CD_OK && st != CD_ERR_NO_MODEL) { ROS_ERROR("%s", cd_last_error(ctx)); return; }
    pcl_msgs::ModelCoefficients ros_coefficients;
    ros_coefficients.header = input->header;
    if (st == CD_OK) ros_coefficients.values.assign(coeff, coeff + 4);      // PCL leaves them empty when no plane was found
    coef_pub.publish(ros_coefficients);
    out.data.resize((size_t)kept * out.point_step);
    out.width = (uint32_t)kept;
    out.row_step = out.width * out.point_step;
    pcl_pub.publish(out);
}

int main(int argc, char** argv) {
    ros::init(argc, argv, "ground_plane_segmentation");
    ros::NodeHandle nh("~");
    nh.param<bool>("invert", invert, true);
    nh.param<double>("voxel_size", voxel_size, 0.01);
    nh.param<double>("distance_threshold", distance_threshold, 0.01);
    nh.param<std::string>("input", input_topic, "/camera/depth/color/points");
    nh.param<std::string>("output", output_topic, "/ground_plane_segmentation/points");
    nh.param<std::string>("plane_coefficients", coefficients_topic, "/ground_plane_segmentation/coefficients");
    ros::Subscriber sub = nh.subscribe(input_topic, 1, callback);
    pcl_pub = nh.advertise<sensor_msgs::PointCloud2>(output_topic, 1);
    coef_pub = nh.advertise<pcl_msgs::ModelCoefficients>(coefficients_topic, 1);
    ros::spin();
    return 0;
}
#else
int main() { return 0; }   // ROS is not available in this build environment
#endif
