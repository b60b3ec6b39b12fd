"""The package the reference's drivers import their model from (``from models.models import ModelsFactory``,
train_ddp.py:7; the factory itself: models/__init__.py:1-15).  One model lives on the accelerated path."""
import importlib

# model_name -> (module, class); resolved on first use so that importing the package does not need the GPU
_MODELS = {'trainer': ('.trainer', 'Trainer')}


class ModelsFactory(object):
    @staticmethod
    def get_by_name(model_name, *args, **kwargs):
        try:
            module, cls = _MODELS[model_name]
        except KeyError:
            raise ValueError('unknown model %r (have: %s)' % (model_name, ', '.join(sorted(_MODELS)))) from None
        model = getattr(importlib.import_module(module, __name__), cls)(*args, **kwargs)
        print('Model %s was created' % model.name)
        return model
