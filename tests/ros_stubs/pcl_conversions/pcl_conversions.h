#pragma once
// the shims hand the PointCloud2 blob to the C-ABI as it is; nothing of pcl_conversions is called
