"""Stub of ``pybullet_data`` (only ``getDataPath`` is used, robot_gym_env.py:269)."""


def getDataPath():
    return '/nonexistent/pybullet_data'
