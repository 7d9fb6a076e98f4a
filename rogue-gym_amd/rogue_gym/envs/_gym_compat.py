"""`gym` is not installed in this image: use it when present, otherwise a minimal stand-in for the
three things the reference's wrappers touch (gym.Env, gym.Wrapper, spaces.Discrete / spaces.Box)."""
try:  # pragma: no cover
    import gym
    from gym import Env, Wrapper, spaces

    Discrete, Box = spaces.discrete.Discrete, spaces.box.Box
    HAVE_GYM = True
except Exception:  # ImportError and broken installs alike
    import numpy as _np

    gym = None
    HAVE_GYM = False

    class Env:
        metadata = {}

        @property
        def unwrapped(self):
            return self

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env
            self.action_space = getattr(env, "action_space", None)
            self.observation_space = getattr(env, "observation_space", None)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

        def __eq__(self, other):
            return isinstance(other, Discrete) and self.n == other.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

        def sample(self):
            return int(_np.random.randint(self.n))

    class Box:
        def __init__(self, low, high, shape, dtype=_np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), _np.dtype(dtype)

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape and self.low == other.low and self.high == other.high
                    and self.dtype == other.dtype)

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low, self.high, self.shape, self.dtype)
