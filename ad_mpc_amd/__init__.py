"""ad_mpc_amd -- MI355X-native batched solve engine for the AD-MPC inner loop of HMCL-UNIST/AD_MPC.

Pure-host modules (config, host, scenarios, dist) import without a GPU; the solver surface
(engine, ocp_solver, ad_3d_optimizer, ad_3d_mpc, create_ros_ad_mpc) needs libadmpc.so and a HIP device
and fails loudly otherwise -- there is no CPU fallback.
"""
__version__ = "0.1.0"
