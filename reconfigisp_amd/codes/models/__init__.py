"""create_model(opt) - mirror of the reference's models/__init__.py:5-22."""
import logging

logger = logging.getLogger('base')


def create_model(opt):
    kind = opt['model']
    if kind == 'darts':
        from .darts_model import DartsModel as M
    elif kind == 'isp':
        from .isp_model import IspModel as M
    elif kind in ('darts_yolo', 'isp_yolo', 'darts_ft'):
        raise NotImplementedError(
            'Model [{:s}] is outside the hot-path scope of this build (YOLO task loss / proxy fine-tuning, '
            'SURVEY.md section 2 rows 16-17).'.format(kind))
    else:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(kind))
    m = M(opt)
    logger.info('Model [{:s}] is created.'.format(kind))
    return m
