class LightningLoggerBase(object):
    def __init__(self, *args, **kwargs):
        pass
