__path__ = __import__("pkgutil").extend_path(__path__, __name__)  # a pip-installed distribution of the same name supplies every module this namespace does not define
