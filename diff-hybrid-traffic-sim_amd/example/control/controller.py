"""Signal controller network (reference example/control/controller.py:3-35): an MLP, observation -> one logit per
action entry, tanh between the hidden layers.  The trainer squashes the logits into the action box."""
import torch as th


class Controller(th.nn.Module):

    def __init__(self, input_size, output_size, network_size=(256, 256)):
        super().__init__()
        widths = [int(input_size)] + [int(w) for w in network_size]
        assert len(widths) > 1, "at least one hidden layer"
        stack = []
        for fan_in, fan_out in zip(widths[:-1], widths[1:]):
            stack += [th.nn.Linear(fan_in, fan_out), th.nn.Tanh()]
        stack.append(th.nn.Linear(widths[-1], int(output_size)))
        self.network = th.nn.Sequential(*stack)         # attribute name = the reference's state-dict prefix ("network.0.weight" ...)

    def forward(self, obs):
        return self.network(obs)
