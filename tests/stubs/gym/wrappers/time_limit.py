from ..core import Wrapper


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None):
        Wrapper.__init__(self, env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._max_episode_steps is not None and self._elapsed_steps >= self._max_episode_steps:
            done = True
        return obs, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)
