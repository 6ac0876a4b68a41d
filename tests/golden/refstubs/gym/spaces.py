from paintrl_amd.spaces import Box, Discrete  # noqa: F401
