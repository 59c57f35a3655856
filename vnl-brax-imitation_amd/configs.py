"""Default hyper-parameters (plain dicts; no hydra).

Values are those of the reference's configs/env_config.yaml:26-106 (rodent env args)
and configs/train_config.yaml:1-17 plus the constants hard-coded at reference
train.py:114-134.
"""

RODENT_ENV_ARGS = dict(
    mjcf_path="./assets/rodent.xml",
    scale_factor=0.9,
    solver="cg",
    iterations=6,
    ls_iterations=6,
    clip_length=250,
    sub_clip_length=10,
    ref_traj_length=5,
    termination_threshold=5,
    end_eff_names=["foot_L", "foot_R", "hand_L", "hand_R"],
    appendage_names=["foot_L", "foot_R", "hand_L", "hand_R", "skull"],
    walker_body_names=[
        "torso", "pelvis", "upper_leg_L", "lower_leg_L", "foot_L", "upper_leg_R", "lower_leg_R", "foot_R",
        "skull", "jaw", "scapula_L", "upper_arm_L", "lower_arm_L", "finger_L", "scapula_R", "upper_arm_R",
        "lower_arm_R", "finger_R",
    ],
    joint_names=[
        "vertebra_1_extend", "hip_L_supinate", "hip_L_abduct", "hip_L_extend", "knee_L", "ankle_L", "toe_L",
        "hip_R_supinate", "hip_R_abduct", "hip_R_extend", "knee_R", "ankle_R", "toe_R", "vertebra_C11_extend",
        "vertebra_cervical_1_bend", "vertebra_axis_twist", "atlas", "mandible", "scapula_L_supinate",
        "scapula_L_abduct", "scapula_L_extend", "shoulder_L", "shoulder_sup_L", "elbow_L", "wrist_L",
        "scapula_R_supinate", "scapula_R_abduct", "scapula_R_extend", "shoulder_R", "shoulder_sup_R", "elbow_R",
        "wrist_R", "finger_R",
    ],
    center_of_mass="torso",
)

TRAIN_CONFIG = dict(
    num_envs=128,
    num_timesteps=3_000_000_000,
    eval_every=10_000,
    episode_length=150,
    batch_size=32,
    learning_rate=6e-4,
    num_minibatches=32,
    num_updates_per_batch=16,
    clipping_epsilon=0.2,
    kl_weight=1e-4,
    intention_latent_size=64,
    encoder_layer_sizes=(256, 128),
    decoder_layer_sizes=(128, 256),
    # hard-coded at reference train.py:114-134
    reward_scaling=1.0,
    normalize_observations=True,
    action_repeat=1,
    unroll_length=20,
    discounting=0.99,
    entropy_cost=1e-3,
    seed=0,
)
