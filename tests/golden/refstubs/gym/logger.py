def _p(*a, **k):
    pass


debug = info = warn = error = _p
