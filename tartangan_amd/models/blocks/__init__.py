from .attention import SelfAttention2d  # noqa
from .discriminator import (  # noqa
    DiscriminatorInput, DiscriminatorOutput, IQNDiscriminatorOutput, ResidualDiscriminatorBlock,
)
from .generator import (  # noqa
    GeneratorInputMLP, GeneratorOutput, ResidualGeneratorBlock, TiledZGeneratorInput,
)
