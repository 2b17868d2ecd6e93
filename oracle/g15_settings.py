"""
Settings of fixture g15_sequence (TEST INFRASTRUCTURE: read by oracle/gen_golden.py, which drives the reference's loop with them on the CPU,
and by tests/test_gpu_slam.py, which runs the HIP drivers with the same numbers).  A sequence small enough for the CPU oracle, long enough
to cross the rules of Mapper.optimize_mapping: joint_opt from the fifth keyframe (src/Mapper.py:519), the extra rays of the newest frames
beyond 20 keyframes (:385-393).  Every frame is tracked, mapped and kept as a keyframe.
"""
G15 = dict(n_frames=34, H=48, W=64, fov_deg=80.0, tex_freq=4.0, room_bound=[[-0.5, 6.5], [-1.1, 3.5], [-1.7, 1.5]], voxel=0.04, log2T=(14, 15),
           tracking=dict(pixels=200, iters=10, lr_T=0.002, lr_R=0.001, ignore_edge_H=4, ignore_edge_W=4, const_speed_assumption=True,
                         activated_mapping_mode=False, uncertainty_ts=0.001, w_color=5, w_depth=1, w_sdf_fs=10, w_sdf_center=200, w_sdf_tail=50),
           mapping=dict(pixels=400, iters=8, iters_first=50, every_frame=1, keyframe_every=1, lr_first_factor=5, lr_factor=1, joint_opt=True,
                        joint_opt_cam_lr=0.001, mapping_window_size=20, LC=True, decoders_lr=0.001, hash_grids_lr=0.05, c_hash_grids_lr=0.05,
                        w_color=5, w_depth=0.1, w_sdf_fs=5, w_sdf_center=200, w_sdf_tail=10),
           n_stratified=32, n_importance=8, truncation=0.06, seed=1500)

# Fixture g16_policy (r6): the same scene under a NON-TRIVIAL policy -- a mapped frame every 3rd frame, a keyframe every 2nd of those
# (src/Mapper.py:487-493,514), activated mapping ON with a threshold the sequence crosses on many frames (uncertainty-triggered doubling
# of the tracking / mapping iterations and the tracking-back flag, src/Tracker.py:352-363; tracking-back frames are mapped and kept as
# keyframes whatever their index, src/Mapper.py:487,514), and a mapping window of 4 so that keyframe_selection_LC's tracking-back branch
# (src/Mapper.py:253-272: the 3 best-overlapping keyframes) has to choose.  Tables of 2^10 / 2^11 entries per level, so that the loop's
# WHOLE state (tables, decoders, keyframe poses, generator state, iteration counts, flags) can be stored at the start of the frames listed
# in `snapshots` (0.4 MB each): a replay can start from the reference's own state, one frame of the loop at a time.  The snapshot frames'
# uncertainty evaluations are all more than 25 % away from the threshold.
G16 = dict(G15, n_frames=30, seed=1700, log2T=(10, 11), snapshots=(12, 16, 18, 22),
           tracking=dict(G15["tracking"], activated_mapping_mode=True, uncertainty_ts=0.0075),
           mapping=dict(G15["mapping"], every_frame=3, keyframe_every=2, mapping_window_size=4))
