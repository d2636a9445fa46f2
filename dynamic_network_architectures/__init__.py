"""Name alias so that plans.json files written by the reference's planner resolve unchanged:
`pydoc.locate("dynamic_network_architectures.architectures.unet.PlainConvUNet")`
(/root/reference/nnunetv2/utilities/get_network_from_plans.py:27) returns the MI355X-native class when this repository
precedes a pip-installed `dynamic_network_architectures` on sys.path.  Only the one class of the hot path is aliased."""
__path__ = __import__("pkgutil").extend_path(__path__, __name__)  # a pip-installed distribution of the same name supplies every module this namespace does not define
