from ship_sim_gym_amd.curriculum import Curriculum  # noqa: F401
