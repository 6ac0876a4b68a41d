from paintrl_amd.spaces import np_random  # noqa: F401
