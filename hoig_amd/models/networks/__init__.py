"""Network registry behind the reference's ``NetworksFactory.get_by_name`` call (models/networks/__init__.py:9-36)."""
import importlib

_GENERATORS = ('generator_base', 'generator_spade', 'generator_spade_attn', 'generator_spade_attn_tiny')


def _build(network_name, args, kwargs):
    if network_name in _GENERATORS:            # one Generator class; the name selects the variant (schema.GeneratorConfig)
        cls = importlib.import_module('.generator', __name__).Generator
        return cls(*args, gen_name=network_name, **kwargs)
    if network_name == 'discriminator_patch_gan':
        return importlib.import_module('.discriminator', __name__).PatchDiscriminator(*args, **kwargs)
    raise ValueError('unknown network %r (have: %s, discriminator_patch_gan)' % (network_name, ', '.join(_GENERATORS)))


class NetworksFactory(object):
    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        network = _build(network_name, args, kwargs)
        print('Network %s was created' % network_name)
        return network
