"""NetworksFactory with the reference's names and call convention (models/networks/__init__.py:9-36)."""


class NetworksFactory(object):
    def __init__(self):
        pass

    @staticmethod
    def get_by_name(network_name, *args, **kwargs):
        if network_name in ('generator_base', 'generator_spade', 'generator_spade_attn', 'generator_spade_attn_tiny'):
            from .generator import Generator
            network = Generator(*args, gen_name=network_name, **kwargs)
        elif network_name == 'discriminator_patch_gan':
            from .discriminator import PatchDiscriminator
            network = PatchDiscriminator(*args, **kwargs)
        else:
            raise ValueError("Network %s not recognized." % network_name)
        print("Network %s was created" % network_name)
        return network
