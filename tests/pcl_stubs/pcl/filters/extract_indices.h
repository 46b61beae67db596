#include <pcl/pcl_stub_core.h>
