"""Stub of ``termcolor`` (param_test_env.py:5 uses ``colored`` for ASCII output only)."""


def colored(text, *args, **kwargs):
    return text
