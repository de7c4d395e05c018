"""Solo12 flat-terrain CaT task: constraint and curriculum configuration.

The ``ConstraintsCfg`` / ``CurriculumCfg`` below carry the reference's values
(solo12/cat_flat_env_cfg.py:259-355 and :383-451: 13 terms / 78 columns, 8 annealed terms).
Scene, physics, rewards, events and observation *terms* of the reference need Isaac Sim /
PhysX (NVIDIA only) and are out of scope; ``SyntheticCfg`` configures the device-resident
synthetic stream that stands in for the simulator.
"""
from cat_envs.shim import CurriculumTermCfg as CurrTerm
from cat_envs.shim import SceneEntityCfg, configclass
from cat_envs.tasks.utils.cat import constraints, curriculums
from cat_envs.tasks.utils.cat.manager_constraint_cfg import ConstraintTermCfg as ConstraintTerm

ALL_JOINTS = [".*_HAA", ".*_HFE", ".*_KFE"]


@configclass
class ConstraintsCfg:
    # Safety soft constraints
    joint_torque = ConstraintTerm(func=constraints.joint_torque, max_p=0.25,
                                  params={"limit": 3.0, "asset_cfg": SceneEntityCfg("robot", joint_names=ALL_JOINTS)})
    joint_velocity = ConstraintTerm(func=constraints.joint_velocity, max_p=0.25,
                                    params={"limit": 16.0, "asset_cfg": SceneEntityCfg("robot", joint_names=ALL_JOINTS)})
    joint_acceleration = ConstraintTerm(func=constraints.joint_acceleration, max_p=0.25,
                                        params={"limit": 800.0,
                                                "asset_cfg": SceneEntityCfg("robot", joint_names=ALL_JOINTS)})
    action_rate = ConstraintTerm(func=constraints.action_rate, max_p=0.25,
                                 params={"limit": 80.0, "asset_cfg": SceneEntityCfg("robot", joint_names=ALL_JOINTS)})
    # Safety hard constraints (knee / base contacts, foot force, front HFE range, roll-over)
    contact = ConstraintTerm(func=constraints.contact, max_p=1.0,
                             params={"asset_cfg": SceneEntityCfg("contact_forces",
                                                                 body_names=["base_link", ".*_UPPER_LEG"])})
    foot_contact_force = ConstraintTerm(func=constraints.foot_contact_force, max_p=1.0,
                                        params={"limit": 50.0,
                                                "asset_cfg": SceneEntityCfg("contact_forces", body_names=".*_FOOT")})
    front_hfe_position = ConstraintTerm(func=constraints.joint_position, max_p=1.0,
                                        params={"limit": 1.3,
                                                "asset_cfg": SceneEntityCfg("robot", joint_names=["FL_HFE", "FR_HFE"])})
    upsidedown = ConstraintTerm(func=constraints.upsidedown, max_p=1.0,
                                params={"limit": 0.0, "asset_cfg": SceneEntityCfg("robot")})
    # Style constraints
    hip_position = ConstraintTerm(func=constraints.joint_position_when_moving_forward, max_p=0.25,
                                  params={"limit": 0.2, "velocity_deadzone": 0.1,
                                          "asset_cfg": SceneEntityCfg("robot", joint_names=[".*_HAA"])})
    base_orientation = ConstraintTerm(func=constraints.base_orientation, max_p=0.25,
                                      params={"limit": 0.1, "asset_cfg": SceneEntityCfg("robot")})
    air_time = ConstraintTerm(func=constraints.air_time, max_p=0.25,
                              params={"limit": 0.25, "velocity_deadzone": 0.1,
                                      "asset_cfg": SceneEntityCfg("contact_forces", body_names=".*_FOOT")})
    no_move = ConstraintTerm(func=constraints.no_move, max_p=0.1,
                             params={"velocity_deadzone": 0.1, "joint_vel_limit": 4.0,
                                     "asset_cfg": SceneEntityCfg("robot", joint_names=ALL_JOINTS)})
    two_foot_contact = ConstraintTerm(func=constraints.n_foot_contact, max_p=0.25,
                                      params={"number_of_desired_feet": 2, "min_command_value": 0.5,
                                              "asset_cfg": SceneEntityCfg("contact_forces", body_names=".*_FOOT")})


@configclass
class SixConstraintsCfg:
    """BASELINE.json config 2: six terms, 42 columns (C3, C4, C7, C13, C12, C8)."""
    joint_torque = ConstraintsCfg.__dataclass_fields__["joint_torque"].default_factory()
    joint_velocity = ConstraintsCfg.__dataclass_fields__["joint_velocity"].default_factory()
    contact = ConstraintsCfg.__dataclass_fields__["contact"].default_factory()
    foot_contact_force = ConstraintsCfg.__dataclass_fields__["foot_contact_force"].default_factory()
    action_rate = ConstraintsCfg.__dataclass_fields__["action_rate"].default_factory()
    base_orientation = ConstraintsCfg.__dataclass_fields__["base_orientation"].default_factory()


@configclass
class TwoConstraintsCfg:
    """BASELINE.json config 1 (plumbing case): two terms, 13 columns - one soft (C3 joint_torque), one hard
    boolean (C7 contact)."""
    joint_torque = ConstraintsCfg.__dataclass_fields__["joint_torque"].default_factory()
    contact = ConstraintsCfg.__dataclass_fields__["contact"].default_factory()


MAX_CURRICULUM_ITERATIONS = 1000


def _anneal(term_name: str) -> CurrTerm:
    return CurrTerm(func=curriculums.modify_constraint_p,
                    params={"term_name": term_name, "num_steps": 24 * MAX_CURRICULUM_ITERATIONS, "init_max_p": 0.25})


@configclass
class CurriculumCfg:
    # soft safety constraints
    joint_torque = _anneal("joint_torque")
    joint_velocity = _anneal("joint_velocity")
    joint_acceleration = _anneal("joint_acceleration")
    action_rate = _anneal("action_rate")
    # style constraints
    hip_position = _anneal("hip_position")
    base_orientation = _anneal("base_orientation")
    air_time = _anneal("air_time")
    two_foot_contact = _anneal("two_foot_contact")


@configclass
class SixCurriculumCfg:
    joint_torque = _anneal("joint_torque")
    joint_velocity = _anneal("joint_velocity")
    action_rate = _anneal("action_rate")
    base_orientation = _anneal("base_orientation")


@configclass
class TwoCurriculumCfg:
    joint_torque = _anneal("joint_torque")


@configclass
class SceneCfg:
    num_envs: int = 4096
    env_spacing: float = 3.0


@configclass
class SimCfg:
    dt: float = 0.005
    device: str = "cuda:0"


@configclass
class SyntheticCfg:
    """stand-in for Isaac Sim: seeded streams of sim state / reward / resets / observations"""
    obs_dim: int = 45            # ang_vel 3 + cmd 3 + gravity 3 + joint_pos 12 + joint_vel 12 + last_action 12
    stream_steps: int = 48       # stream length (cycled)
    seed_offset: int = 1234
    exact_reset_sync: bool = False


@configclass
class Solo12FlatEnvCfg:
    scene: SceneCfg = SceneCfg(num_envs=4096, env_spacing=3.0)
    sim: SimCfg = SimCfg()
    constraints: ConstraintsCfg = ConstraintsCfg()
    curriculum: CurriculumCfg = CurriculumCfg()
    synthetic: SyntheticCfg = SyntheticCfg()
    decimation: int = 4
    episode_length_s: float = 10.0
    seed: int = 42


@configclass
class Solo12FlatEnvCfg_PLAY(Solo12FlatEnvCfg):
    scene: SceneCfg = SceneCfg(num_envs=50, env_spacing=3.0)
    curriculum: object = None
