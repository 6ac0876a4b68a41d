class Error(Exception):
    pass


class InvalidFrame(Error):
    pass


class DependencyNotInstalled(Error):
    pass
