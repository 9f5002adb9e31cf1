import gymnasium as gym


class MultiAgentEnv(gym.Env):
    def __init__(self):
        pass
