from ship_sim_gym_amd.game_map import gen_river_poly  # noqa: F401
