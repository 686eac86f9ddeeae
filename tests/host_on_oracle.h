/* Test-only: force-included (-include) when tests/Makefile builds the host layer on the CPU oracle, whose
 * exports carry the hope_oracle_ prefix.  Never used by the product build (hope_amd/host/Makefile). */
#define hope_config_default hope_oracle_config_default
#define hope_create hope_oracle_create
#define hope_destroy hope_oracle_destroy
#define hope_last_error hope_oracle_last_error
#define hope_set_pose hope_oracle_set_pose
#define hope_set_frame_poses hope_oracle_set_frame_poses
#define hope_set_height_window hope_oracle_set_height_window
#define hope_segment_depth_u16 hope_oracle_segment_depth_u16
#define hope_segment_cloud_xyz hope_oracle_segment_cloud_xyz
#define hope_frame_counts hope_oracle_frame_counts
#define hope_get_downsampled hope_oracle_get_downsampled
#define hope_get_fit_indices hope_oracle_get_fit_indices
#define hope_get_clusters hope_oracle_get_clusters
#define hope_get_cluster_indices hope_oracle_get_cluster_indices
#define hope_get_max_plane hope_oracle_get_max_plane
#define hope_last_timing hope_oracle_last_timing
#define hope_clusters_upon_plane hope_oracle_clusters_upon_plane
#define hope_get_upon_cluster_sizes hope_oracle_get_upon_cluster_sizes
#define hope_get_upon_cluster hope_oracle_get_upon_cluster
