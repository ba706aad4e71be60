"""Model factory - the ``opt['model']`` names of the reference (models/__init__.py:5-22)."""
import importlib
import logging

_BUILT = {'darts': ('.darts_model', 'DartsModel'), 'isp': ('.isp_model', 'IspModel'),
          'darts_ft': ('.darts_ft_model', 'DartsFtModel')}
_OUT_OF_SCOPE = {'darts_yolo': 'YOLOv3 task loss', 'isp_yolo': 'YOLOv3 task loss'}


def create_model(opt):
    name = opt['model']
    if name in _OUT_OF_SCOPE:
        raise NotImplementedError('Model [{:s}] ({}) is outside the hot-path scope of this build '
                                  '(SURVEY.md section 2 rows 16-17).'.format(name, _OUT_OF_SCOPE[name]))
    if name not in _BUILT:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(name))
    module, cls = _BUILT[name]
    model = getattr(importlib.import_module(module, __name__), cls)(opt)
    logging.getLogger('base').info('Model [{:s}] is created.'.format(name))
    return model
