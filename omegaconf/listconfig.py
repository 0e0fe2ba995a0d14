"""`omegaconf.listconfig.ListConfig` (imported by the reference's UNetModel, openaimodel.py:595): a list."""


class ListConfig(list):
    pass
