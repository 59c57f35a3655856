"""ppo_imitation: the PPO + intention-network side of the hot path (reference ppo_imitation/*)."""
