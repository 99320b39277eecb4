from enum import Enum, auto


class TrainingCallbackLocation(Enum):
    BEFORE_TRAIN_ITERATION = auto()
    AFTER_TRAIN_ITERATION = auto()
    AFTER_TRAIN = auto()


class TrainingCallbackAttributes:
    def __init__(self, optimizers, grad_scaler, pipeline):
        self.optimizers, self.grad_scaler, self.pipeline = optimizers, grad_scaler, pipeline


class TrainingCallback:
    def __init__(self, where_to_run, func, update_every_num_iters=None, args=None):
        self.where_to_run, self.func, self.every, self.args = where_to_run, func, update_every_num_iters, args or []

    def run_callback_at_location(self, step, location):
        if location in self.where_to_run and (self.every is None or step % self.every == 0):
            self.func(*self.args, step=step)
