"""Learning-rate schedule of the reference driver (main_frame_val_text_missing.py:318-321):
linear warm-up over 5 epochs, then x0.9 every 10 epochs; stepped once per epoch (:342)."""


def warm_up_with_step_lr(epoch, warm_up_epochs=5, gamma=0.9, stepsize=10):
    if epoch < warm_up_epochs:
        return (epoch + 1) / warm_up_epochs
    return gamma ** ((epoch + 1 - warm_up_epochs) // stepsize)


def lr_at(epoch, base_lr=1e-4):
    """Learning rate in effect during `epoch` (0-based), = what LambdaLR sets after `epoch` scheduler steps."""
    return base_lr * warm_up_with_step_lr(epoch)
