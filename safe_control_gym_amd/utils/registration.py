"""Registry behind ``make()`` -- the drop-in door of the engine.

Same call surface as the reference's ``safe_control_gym/utils/registration.py:138-161``
(``register(id, entry_point, config_entry_point)``, ``make(id, *args, **kwargs)``, ``spec(id)``,
``get_config(id)``; ``make`` attaches ``.spec`` to the object it builds, :81-86).  The entry points
registered here are the batched, device-resident environments of this package.
"""
import copy
import importlib
import os

import yaml

_CONFIG_GROUPS = ('timing', 'vehicle', 'start', 'objective', 'actions', 'safety', 'perturbations', 'interface', 'engine')

_SPECS = {}


class Spec:
    """What ``register`` stores for one id (registration.py:25-86)."""

    def __init__(self, id, entry_point=None, config_entry_point=None):
        self.id = id
        self.entry_point = entry_point
        self.config_entry_point = config_entry_point

    def __repr__(self):
        return 'Spec({})'.format(self.id)

    def _resolve(self, name):
        module, attr = name.split(':')
        return getattr(importlib.import_module(module), attr)

    def get_config(self):
        cep = self.config_entry_point
        if cep is None:
            return {}
        if not isinstance(cep, str):
            raise Exception('Config type {} is not supported.'.format(cep))
        if cep.endswith('.yaml'):
            module, fname = cep.split(':')
            folder = os.path.dirname(importlib.import_module(module).__file__)
            with open(os.path.join(folder, fname)) as f:
                groups = yaml.safe_load(f)
            flat = {}
            for name, group in groups.items():           # the defaults files group the options; callers get them flat
                if isinstance(group, dict) and name in _CONFIG_GROUPS:
                    flat.update(group)
                else:
                    flat[name] = group
            return flat
        return self._resolve(cep)

    def make(self, *args, **kwargs):
        if self.entry_point is None:
            raise Exception('Attempting to make deprecated env {}.'.format(self.id))
        factory = self.entry_point if callable(self.entry_point) else self._resolve(self.entry_point)
        obj = factory(*args, **kwargs)
        tag = copy.deepcopy(self)
        if hasattr(obj, 'spec'):
            obj.instance_spec = tag
        else:
            obj.spec = tag
        return obj


def register(id, **kwargs):
    if id in _SPECS:
        raise Exception('Cannot re-register id: {}'.format(id))
    _SPECS[id] = Spec(id, **kwargs)


def spec(id):
    if ':' in id:
        module, id = id.split(':')
        try:
            importlib.import_module(module)
        except ImportError:
            raise Exception('A module ({}) was specified for the environment but was not found'.format(module))
    try:
        return _SPECS[id]
    except KeyError:
        raise Exception('Key not found in registry.')


def make(id, *args, **kwargs):
    return spec(id).make(*args, **kwargs)


def load(name):
    """``'module:attr'`` -> the attribute (registration.py:12-22)."""
    module, attr = name.split(':')
    return getattr(importlib.import_module(module), attr)


class Registry:
    """The reference's registry object (registration.py:89-135) as a view of this module's table."""

    @property
    def specs(self):
        return _SPECS

    def make(self, id, *args, **kwargs):
        return make(id, *args, **kwargs)

    def all(self):
        return all_specs()

    def spec(self, id):
        return spec(id)

    def register(self, id, **kwargs):
        return register(id, **kwargs)

    def get_config(self, id):
        return get_config(id)


registry = Registry()


def get_config(id):
    return spec(id).get_config()


def all_specs():
    return _SPECS.values()
